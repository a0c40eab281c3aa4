// Composite renders for gfx950: the per-sub-tile list of "A and B rendered together" as the MERGE of the sorted lists of two
// renders A and B that already exist -- no preprocess, no cell scatter, no sub-tile binning, no sort for the composite.
//
// Why: ExAvatar renders five images per training sample with ONE camera (reference avatar/main/model.py:119-167):
//     scene, human, cat(scene.detach(), human), human_refined, cat(scene.detach(), human_refined)
// ("shared preprocess / sort for the 5 renders ... biggest real-world lever on ExAvatar iters/s").  The two
// composites contain exactly the Gaussians of two renders of the same batch, seen by the same camera: their splat records
// are the sources' records, and the depth-sorted list of a sub-tile is the merge of the sources' two sorted lists (order of
// the concatenated render: ascending depth bits, ties by index -- every A (scene) index precedes every B (human) index, so A
// wins ties).  Two small launches replace five:
// Where the human is absent the composite IS the scene render: with source A's finished images at hand (and equal
// backgrounds) a sub-tile without B entries gets an empty list and its pixels are copied from A's images -- in ExAvatar's
// composites ~3/4 of the image skip merge, blend and every backward wave (round 4; per five-render iteration: composite
// forward 136 -> see DESIGN.md, bit-identical).
//   compose_kernel   block 0 (one workgroup, all range loads of a thread in flight at once): list lengths nA + nB -> 64-aligned
//                    ranges of the composite's own instance space, header; the other blocks zero-fill the composite's owner /
//                    blended-mask / touched arrays and copy the launch order of the blend from the heavier source
//   merge_kernel     one wave per sub-tile: 64 candidates of each source per trip (sorted 64-bit keys, kept by the sources'
//                    sort: RenderFwdArgs.keep_sorted_keys), each candidate's output position = its own index + its rank in
//                    the other window (binary search in LDS: lower bound for A, upper bound for B), ids written with the
//                    source in bit 31 (SRC_B), batch owners written alongside
// then render_fwd_kernel<STORE, TWO = true> / render_bwd_kernel<.., PREFIX = true> (two record arrays) and the unchanged
// preprocess_bwd_kernel on B's records with the composite's partial sums: A is a constant of the backward pass (the detached
// scene), like ExaRasterBackwardJob.grad_first.  Images are bit-identical to rendering the concatenation.
#include "common.h"

namespace exa {

constexpr int CBLOCK = 1024;             // compose_kernel (one workgroup scans every sub-tile: all loads of a thread in flight at once)
constexpr int MBLOCK = 256;              // merge_kernel: four independent waves
constexpr int C_ZERO_WGS = 32;
constexpr int C_PER = 16;                // sub-tiles per thread and trip of the scan

__device__ __forceinline__ uint32_t list_slots(uint32_t n) { return n ? (n + BATCH - 1) / BATCH + 1 : 0u; }   // batches + end slot

__global__ __launch_bounds__(CBLOCK) void compose_kernel(Batch<ComposeArgs> batch) {
    __shared__ uint32_t s_wave[CBLOCK / 64];
    const ComposeArgs& a = batch.v[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int subtiles = a.grid.subtiles;
    if (blockIdx.x > 0) {
        // ---- zero-filled section of the composite's workspace: owners (merge_kernel), blended masks (render_fwd), touched (render_bwd)
        const size_t n16 = compose_zero_bytes(a.capacity, a.capacity_b) / 16;
        uint4* p = a.bw.owner;
        const size_t first = (size_t)((int)blockIdx.x - 1) * CBLOCK + tid, stride = (size_t)C_ZERO_WGS * CBLOCK;
        for (size_t i = first; i < n16; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
        // the composite's radii / is_vis = the sources' one behind the other (ExaRasterComposeJob.radii_out)
        const size_t n_ab = (size_t)a.P_a + (size_t)a.P_b;
        if (a.radii_out)
            for (size_t i = first; i < n_ab; i += stride) a.radii_out[i] = i < (size_t)a.P_a ? a.radii_a[i] : a.radii_b[i - a.P_a];
        if (a.vis_out)
            for (size_t i = first; i < n_ab; i += stride) a.vis_out[i] = i < (size_t)a.P_a ? a.vis_a[i] : a.vis_b[i - a.P_a];
        return;                                                  // (the launch order of the blend: merge_kernel writes it)
    }
    const uint2* __restrict__ ra = a.tw_a.ranges;
    const uint2* __restrict__ rb = a.tw_b.ranges;
    const bool src_overflow = a.tw_a.header->overflow != 0u || a.tw_b.header->overflow != 0u;
    // A's own pixels stand in where B has no entry (include/exa_raster.h, ExaRasterComposeJob.a_color): such a sub-tile gets
    // an EMPTY list here -- no merge, no batch slots, nothing for the backward to look at -- and render_fwd copies it
    const bool reuse = reuse_a_pixels(a.src_color, a.src_bg, a.bg);
    // ---- 64-aligned ranges: exclusive prefix of the slot counts over the sub-tiles (cell-major order, like the sources) ------
    // A wave takes whole CELLS (lane = sub-tile of the cell: 512-byte coalesced loads and stores; a thread-major split of the
    // sub-tiles touched 64 cache lines per load instruction and cost 25 us on this one CU): per trip 16 waves x 16 cells, the
    // cells' totals meet in LDS, one wave scans them, every wave writes its cells' ranges.
    __shared__ uint32_t s_cell[CBLOCK / 64 * C_PER];
    const int cells = subtiles / SUBS_PER_CELL;
    uint32_t carry = 0;
    for (int cell0 = 0; cell0 < cells; cell0 += CBLOCK / 64 * C_PER) {
        uint32_t len[C_PER], excl[C_PER];
        uint2 x[C_PER], y[C_PER];
#pragma unroll
        for (int i = 0; i < C_PER; ++i) {
            const int st = min(cell0 + wave * C_PER + i, cells - 1) * SUBS_PER_CELL + lane;
            x[i] = ra[st]; y[i] = rb[st];
        }
#pragma unroll
        for (int i = 0; i < C_PER; ++i) {
            const bool valid = cell0 + wave * C_PER + i < cells && !src_overflow;
            len[i] = valid ? (x[i].y - x[i].x) + (y[i].y - y[i].x) : 0u;
            if (reuse && y[i].y == y[i].x) len[i] = 0u;
            const uint32_t ns = list_slots(len[i]);
            uint32_t incl = ns;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d, 64);
                if (lane >= d) incl += o;
            }
            excl[i] = incl - ns;
            if (lane == 63) s_cell[wave * C_PER + i] = incl;
        }
        __syncthreads();
        if (wave == 0) {                                         // exclusive prefix of the 256 cell totals of this trip
            uint32_t v[4], sum = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = s_cell[lane * 4 + q]; sum += v[q]; }
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d, 64);
                if (lane >= d) incl += o;
            }
            uint32_t run = incl - sum;
#pragma unroll
            for (int q = 0; q < 4; ++q) { s_cell[lane * 4 + q] = run; run += v[q]; }
            if (lane == 63) s_wave[0] = incl;                    // total of the trip
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < C_PER; ++i) {
            const int cell = cell0 + wave * C_PER + i;
            if (cell < cells) {
                const uint32_t begin = (carry + s_cell[wave * C_PER + i] + excl[i]) * BATCH;
                a.tw.ranges[cell * SUBS_PER_CELL + lane] = make_uint2(begin, begin + len[i]);
            }
        }
        carry += s_wave[0];
        __syncthreads();                                         // (s_cell / s_wave are rewritten by the next trip)
    }
    const uint32_t total = carry;
    // an overflow can only come from the sources (their lists are empty then: every length above is 0) or from a caller that
    // passed less than capacity_a + capacity_b: empty every list again
    const bool overflow = src_overflow || (uint64_t)total * BATCH > a.capacity;
    if (overflow && !src_overflow) {
        __syncthreads();
        for (int st = tid; st < subtiles; st += CBLOCK) a.tw.ranges[st] = make_uint2(0u, 0u);
    }
    if (tid == 0) {
        ExaRasterHeader* h = a.tw.header;
        h->num_rendered = total * BATCH; h->overflow = overflow ? 1u : 0u; h->max_tile_list = 0u;
        h->num_visible = a.tw_a.header->num_visible + a.tw_b.header->num_visible;
        h->num_instances = a.tw_a.header->num_instances + a.tw_b.header->num_instances;
        h->active_cells = 0u;
        h->num_tile_instances = a.tw_a.header->num_tile_instances + a.tw_b.header->num_tile_instances;
        a.tw.bwd_meta[2] = 0u;                                   // no batch order for the backward of a composite (slot order)
        if (a.host_hdr) {
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            const v4u v = {total * BATCH, overflow ? 1u : 0u, h->num_visible, a.hdr_tag};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(a.host_hdr), "v"(v) : "memory");
        }
    }
}

// One wave per sub-tile, four per workgroup.
__global__ __launch_bounds__(MBLOCK) void merge_kernel(Batch<ComposeArgs> batch) {
    __shared__ uint32_t s_da[MBLOCK / 64][64], s_db[MBLOCK / 64][64], s_out[MBLOCK / 64][64];
    const ComposeArgs& a = batch.v[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int st = (int)blockIdx.x * (MBLOCK / 64) + wave;
    if (st >= a.grid.subtiles) return;
    // ---- launch order of the blend (heavy lists first, common.h): the order of the source with more instances.  In
    // ExAvatar's composites that is the scene, whose lists dominate the merged ones; a histogram of the exact merged
    // lengths cost a 27-88 us single-workgroup pass (two variants measured) for a launch that queues ~16 waves per SIMD
    // anyway.  The records carry the sub-tile only: the blend reads its range from tw.ranges.  Written HERE (record number
    // `st` by the wave of sub-tile `st`) and not by compose_kernel, because the sources' order is a product of their SORT
    // launch, which compose_kernel need not wait for (EXA_RASTER_STAGE_NO_SORT).
    if (lane == 0) {
        const uint4* __restrict__ src = a.tw_a.header->num_instances >= a.tw_b.header->num_instances ? a.tw_a.slots : a.tw_b.slots;
        a.tw.slots[st] = make_uint4(0u, 0u, src[st].z, 0u);
    }
    const uint2 rc = a.tw.ranges[st];
    const int n = (int)(rc.y - rc.x);
    if (n == 0) return;                                          // empty list (or overflow: every range is empty)
    const uint2 xa = a.tw_a.ranges[st], xb = a.tw_b.ranges[st];
    const int nA = (int)(xa.y - xa.x), nB = (int)(xb.y - xb.x);
    const unsigned long long* __restrict__ ka = a.bw_a.keys + xa.x;
    const unsigned long long* __restrict__ kb = a.bw_b.keys + xb.x;
    uint32_t* __restrict__ out = a.bw.sorted + rc.x;
    for (int bq = lane; bq * BATCH < n; bq += 64)               // batch owners: what a backward wave needs to find its work
        a.bw.owner[rc.x / BATCH + bq] = make_uint4((uint32_t)st + 1u, rc.x, (uint32_t)n, 0u);
    uint32_t* da = s_da[wave];
    uint32_t* db = s_db[wave];
    uint32_t* so = s_out[wave];
    int ia = 0, ib = 0;
    for (int base = 0; base < n; base += 64) {
        const bool va = ia + lane < nA, vb = ib + lane < nB;
        const unsigned long long ca = va ? ka[ia + lane] : ~0ull, cb = vb ? kb[ib + lane] : ~0ull;
        const uint32_t dA = (uint32_t)(ca >> 32), dB = (uint32_t)(cb >> 32);       // depth bits (positive floats: bit order = value order)
        da[lane] = dA; db[lane] = dB;
        wave_lds_fence();
        // A candidate: in front of it go its predecessors in A and the B candidates with a strictly smaller depth
        uint32_t cA = 0, cB = 0;
#pragma unroll
        for (int s2 = 32; s2 > 0; s2 >>= 1) {
            if (db[cA + s2 - 1] < dA) cA += s2;
            if (da[cB + s2 - 1] <= dB) cB += s2;                 // B candidate: A candidates with depth <= its own go first
        }
        if (db[cA] < dA) cA += 1;                                // (the loops stop at 63)
        if (da[cB] <= dB) cB += 1;
        const uint32_t pA = (uint32_t)lane + cA, pB = (uint32_t)lane + cB;
        const bool tA = va && pA < 64u, tB = vb && pB < 64u;
        if (tA) so[pA] = (uint32_t)ca;
        if (tB) so[pB] = (uint32_t)cb | SRC_B;
        wave_lds_fence();
        if (base + lane < n) out[base + lane] = so[lane];
        ia += __popcll(__ballot(tA));
        ib += __popcll(__ballot(tB));
        wave_lds_fence();                                        // the next trip overwrites the windows
    }
}

// parts: bit 0 = ranges (compose_kernel: needs the sources' ranges, i.e. their binning), bit 1 = merges (needs their sorts)
hipError_t launch_compose(const ComposeArgs* a, int K, hipStream_t s, int parts) {
    int subtiles = 0;
    for (int k = 0; k < K; ++k) subtiles = subtiles > a[k].grid.subtiles ? subtiles : a[k].grid.subtiles;
    if (subtiles == 0) return hipSuccess;
    const Batch<ComposeArgs> b = make_batch(a, K);
    if (parts & 1) compose_kernel<<<dim3(1 + C_ZERO_WGS, K), CBLOCK, 0, s>>>(b);
    if (parts & 2) merge_kernel<<<dim3((subtiles + MBLOCK / 64 - 1) / (MBLOCK / 64), K), MBLOCK, 0, s>>>(b);
    return hipGetLastError();
}

}  // namespace exa
