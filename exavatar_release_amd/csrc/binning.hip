// Binning for gfx950: the "duplicate then radix sort" of the upstream rasterizer restructured as an
// MSD radix sort with LDS-resident histograms and NO global atomics:
//   digit 1 = 64x64-px cell     per-chunk histogram in preprocess_fwd.hip, stored as a row of the (chunk, cell)
//                               count matrix; col_scan_kernel (column totals + per-chunk offsets),
//                               cell_scan_kernel (prefix over cells), cell_scatter_kernel (entries -> buckets)
//   digit 2 = 8x8-px sub-tile   subtile_count_kernel + subtile_bin_kernel: four workgroups per cell, counts and
//                               ranks in LDS, emits (depth bits << 32 | id) keys grouped by sub-tile
//   digit 3 = depth (+ id)      sorted per sub-tile inside LDS by sort_subtiles_kernel (render_fwd.hip)
// which reproduces the order of upstream's stable global sort on (tile << 32 | depth bits): ascending
// depth, ties by ascending Gaussian id.  Counts travel between workgroups through plain-store matrices and
// kernel boundaries: device-scope atomics run at ~12 G/s on MI355X and serialise per address.
//
// Replaces upstream InclusiveSum + duplicateWithKeys + SortPairs(tile digit) + identifyTileRanges of the
// rasterizer behind reference avatar/common/nets/module.py:632-640.
// HBM traffic: reads 16 B of every visible splat record twice, writes 16 B per cell entry and 8 B per
// instance, 4 B inst_off per Gaussian; scans are O(chunks x cells).
#include <stdlib.h>
#include "common.h"

namespace exa {

constexpr int SCAN_THREADS = 1024;
constexpr int SC_BLOCK_THREADS = 1024;   // = SC_BLOCK (cell_scatter_kernel), declared further down

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// Exclusive scan of one value per thread across the workgroup; `total` receives the workgroup sum.
// s_tmp needs (blockDim / 64) entries.  Ends with a barrier, s_tmp may be reused immediately.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_tmp, uint32_t& total) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int i = 0; i < nw; ++i) {
        const uint32_t ws = s_tmp[i];
        if (i < wave) base += ws;
        tot += ws;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

// Zero `n16` 16-byte words.  Used instead of hipMemsetAsync so that a captured hipGraph contains only
// kernel nodes (memset nodes of the bundled ROCm runtime did not re-zero the buffer on replay).
__global__ __launch_bounds__(BLOCK) void zero_kernel(uint4* __restrict__ p, size_t n16) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0, 0, 0, 0);
}

// Column pass over the (chunk, cell) count matrix written by preprocess_fwd: 32 columns (cells) per workgroup,
// 32 row blocks.  Produces the column totals (cell_cnt) and replaces the low word of every entry by the number of
// entries the EARLIER chunks put into that cell, so that cell_scatter_kernel needs no atomic cursor and places
// entries deterministically (by chunk, then by LDS rank).
constexpr int COL_W = 32, COL_RB = SCAN_THREADS / COL_W;
__global__ __launch_bounds__(SCAN_THREADS) void col_scan_kernel(Batch<BinArgs> batch) {
    __shared__ unsigned long long s_part[COL_RB][COL_W];
    const BinArgs& a = batch.v[blockIdx.y];
    const TileWs& w = a.tw;
    const int cells = a.grid.cells, chunks = a.chunks;
    if ((int)blockIdx.x * COL_W >= cells) return;
    const int tid = threadIdx.x, col = tid & (COL_W - 1), rb = tid / COL_W;
    const int c = blockIdx.x * COL_W + col;
    const int rows_per = (chunks + COL_RB - 1) / COL_RB;
    const int r0 = min(chunks, rb * rows_per), r1 = min(chunks, r0 + rows_per);
    unsigned long long* m = w.chunk_cell + c;
    unsigned long long sum = 0ull;
    if (c < cells)
        for (int r = r0; r < r1; ++r) sum += m[(size_t)r * cells];
    s_part[rb][col] = sum;
    __syncthreads();
    if (c >= cells) return;
    unsigned long long before = 0ull, total = 0ull;
    for (int q = 0; q < COL_RB; ++q) {
        const unsigned long long v = s_part[q][col];
        if (q < rb) before += v;
        total += v;
    }
    if (rb == 0) w.cell_cnt[c] = total;
    uint32_t run = (uint32_t)before;
    for (int r = r0; r < r1; ++r) {
        const unsigned long long v = m[(size_t)r * cells];
        m[(size_t)r * cells] = (v & 0xffffffff00000000ull) | run;
        run += (uint32_t)v;
    }
}

// One workgroup: exclusive prefix of the per-cell (entries, instances) totals and of the per-chunk instance
// counts, the header, the LPT order of the cells.
__device__ __forceinline__ unsigned long long wave_incl_scan64(unsigned long long v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}
// exclusive scan of two 32-bit counters packed in one u64 (no carry between the halves: both sums stay < 2^32)
__device__ __forceinline__ unsigned long long block_excl_scan64(unsigned long long v, unsigned long long* s_tmp,
                                                                unsigned long long& total) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    const unsigned long long incl = wave_incl_scan64(v);
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
    for (int i = 0; i < nw; ++i) {
        const unsigned long long ws = s_tmp[i];
        if (i < wave) base += ws;
        tot += ws;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

// Launch order of the per-cell kernels: cells bucketed by floor(log2(instances)) (33 buckets), heaviest bucket first,
// so the heaviest cells start early and the tail of the launch is light; the empty cells come last and
// header.active_cells counts the others.  One workgroup; `inst_of(c)` returns the instance count of cell c.  Every
// record carries the cell's entry range and first instance slot (cell_off, already written to global memory by this
// workgroup), so that the per-cell workgroups of the next kernels need ONE load instead of a chain of three.
// ADAPTIVE split of the cells over the BIN_PARTS * cells workgroups of the two-launch sub-tile binning (subtile_count /
// subtile_bin below): a cell gets one workgroup per `per` entries (per >= 1024: one trip of their entry loop), at most
// MAX_PARTS, at least one (an empty cell's workgroup publishes its empty ranges); the workgroups beyond the sum find
// NO_PART and leave.  With a fixed four parts per cell those launches lasted as long as the parts of the heaviest cells --
// three trips, 8-10 us against 5 us for a one-trip part (workgroup timeline) -- while a cell of
// fifty entries still occupied four workgroups; eight fixed parts gained 2.5 us, twelve nothing, sixteen lost 4 (the
// dispatch of 4 096 mostly idle 1024-thread workgroups).  Deriving the mapping in every binning workgroup (cell records
// + one block scan) cost more than the balance gained (+1.8 us): the table is written HERE, once, by the one workgroup
// that has just written the cell records.  Cells by rank (heavy first), a cell's parts consecutive.
constexpr int MAX_PARTS = 16;
constexpr uint32_t NO_PART = 0xffffffffu;
#ifndef EXA_PART_ENTRIES
#define EXA_PART_ENTRIES 1024           // = BIN_THREADS: one trip of the entry loops (A/B on C3: 512 the same, 768 / 1536 +0.8-1 us)
#endif
constexpr int PART_ENTRIES = EXA_PART_ENTRIES;
__device__ __forceinline__ void write_part_table(const TileWs& w, int cells) {
    __shared__ uint32_t s_ptmp[32];
    const int tid = threadIdx.x, nthreads = (int)blockDim.x;
    const int slots = cells * BIN_PARTS;
    if (cells > nthreads || cells >= (1 << 12)) return;         // (images that large take the one-workgroup-per-cell kernel)
    __threadfence_block();
    __syncthreads();                                             // cell_desc complete (written by this workgroup)
    const uint32_t entries = w.cell_off[cells].x;
    // sum of the parts <= entries / per + cells <= BIN_PARTS * cells
    const uint32_t denom = (uint32_t)((BIN_PARTS - 1) * cells);
    const uint32_t per = max((uint32_t)PART_ENTRIES, (entries + denom - 1u) / denom);
    const uint4 d = tid < cells ? w.cell_desc[tid] : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t e = d.z - d.y;
    const uint32_t mine = tid < cells ? min((uint32_t)MAX_PARTS, max(1u, (e + per - 1u) / per)) : 0u;
    uint32_t total;
    const uint32_t first = block_excl_scan(mine, s_ptmp, total);
    const uint32_t share = mine ? (e + mine - 1u) / mine : 0u;
    for (uint32_t p = 0; p < mine; ++p) {
        const uint32_t lo = min(d.z, d.y + p * share), hi = min(d.z, lo + share);
        w.part_desc[first + p] = make_uint4(d.x | ((uint32_t)tid << 12) | (p << 24) | ((mine - 1u) << 28), lo, hi, d.w);
    }
    for (int i = (int)total + tid; i < slots; i += nthreads) w.part_desc[i] = make_uint4(NO_PART, 0u, 0u, 0u);
}

template <typename F>
__device__ __forceinline__ void write_cell_order(const TileWs& w, int cells, F inst_of) {
    __shared__ uint32_t s_bucket[34];
    const int tid = threadIdx.x;
    if (tid < 34) s_bucket[tid] = 0u;
    __syncthreads();
    for (int c = tid; c < cells; c += (int)blockDim.x) {
        const uint32_t n = inst_of(c);
        atomicAdd(&s_bucket[n ? 32 - __clz(n) : 0], 1u);       // bucket 0 = empty, 32 = largest
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int b = 32; b >= 0; --b) { const uint32_t n = s_bucket[b]; s_bucket[b] = run; run += n; }
        // the empty cells sit at the end of cell_order: the per-cell kernels that follow stop at this count instead of
        // sending a workgroup through three dependent loads for every empty cell (~75 % of the cells of an avatar view)
        w.header->active_cells = s_bucket[0];
    }
    __syncthreads();
    for (int c = tid; c < cells; c += (int)blockDim.x) {
        const uint32_t n = inst_of(c);
        const uint2 o0 = w.cell_off[c], o1 = w.cell_off[c + 1];
        w.cell_desc[atomicAdd(&s_bucket[n ? 32 - __clz(n) : 0], 1u)] = make_uint4((uint32_t)c, o0.x, o1.x, o0.y);
    }
    write_part_table(w, cells);
}

// Two exclusive scans across a 1024-thread workgroup behind ONE pair of barriers (a packed u64 and a u32 per thread).
__device__ __forceinline__ void block_excl_scan_pair(unsigned long long x, uint32_t y, unsigned long long* s64, uint32_t* s32,
                                                     unsigned long long& x_excl, unsigned long long& x_total,
                                                     uint32_t& y_excl, uint32_t& y_total) {
    constexpr int NW = SC_BLOCK_THREADS / 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long ix = wave_incl_scan64(x);
    const uint32_t iy = wave_incl_scan(y);
    if (lane == 63) { s64[wave] = ix; s32[wave] = iy; }
    __syncthreads();
    unsigned long long bx = 0ull, tx = 0ull;
    uint32_t by = 0u, ty = 0u;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const unsigned long long wx = s64[i];
        const uint32_t wy = s32[i];
        bx += i < wave ? wx : 0ull; tx += wx;
        by += i < wave ? wy : 0u; ty += wy;
    }
    __syncthreads();
    x_excl = bx + ix - x; x_total = tx;
    y_excl = by + iy - y; y_total = ty;
}

__global__ __launch_bounds__(SCAN_THREADS) void cell_scan_kernel(Batch<BinArgs> batch) {
    __shared__ unsigned long long s_tmp[SCAN_THREADS / 64];
    const BinArgs& a = batch.v[blockIdx.y];
    const TileWs& w = a.tw;
    const int cells = a.grid.cells, chunks = a.chunks;
    const int tid = threadIdx.x;
    // every global input of the first trip is requested up front: the kernel is one workgroup of pure latency
    const unsigned long long v0 = tid < cells ? w.cell_cnt[tid] : 0ull;
    const uint32_t ci0 = tid < chunks ? w.chunk_inst[tid] : 0u, cv0 = tid < chunks ? w.chunk_vis[tid] : 0u;
    unsigned long long carry = 0ull;                            // slots * 64 << 32 | entries
    uint32_t true_inst = 0;
    for (int base = 0; base < cells; base += SCAN_THREADS) {
        const int c = base + tid;
        const unsigned long long v = base == 0 ? v0 : (c < cells ? w.cell_cnt[c] : 0ull);
        true_inst += (uint32_t)(v >> 32);
        const unsigned long long packed =
            ((unsigned long long)(cell_slots((uint32_t)(v >> 32)) * BATCH) << 32) | (uint32_t)v;   // instance space in 64-slots
        unsigned long long tot;
        const unsigned long long x = carry + block_excl_scan64(packed, s_tmp, tot);
        if (c < cells) w.cell_off[c] = make_uint2((uint32_t)x, (uint32_t)(x >> 32));
        carry += tot;
    }
    // prefix of the per-chunk instance counts; totals of true instances and visible Gaussians ride along
    unsigned long long ccarry = 0ull, totals = 0ull;
    uint32_t vis = 0;
    for (int base = 0; base < chunks || base == 0; base += SCAN_THREADS) {
        const int c = base + tid;
        const uint32_t v = base == 0 ? ci0 : (c < chunks ? w.chunk_inst[c] : 0u);
        vis += base == 0 ? cv0 : (c < chunks ? w.chunk_vis[c] : 0u);
        const bool last = base + SCAN_THREADS >= chunks;
        // high word: on the last trip the per-thread (true_inst, vis) partial sums are folded in as a second scan
        unsigned long long tot;
        const unsigned long long x = block_excl_scan64((unsigned long long)v | ((unsigned long long)(last ? vis : 0u) << 32), s_tmp, tot);
        if (c < chunks) w.chunk_off[c] = (uint32_t)ccarry + (uint32_t)x;
        ccarry += tot & 0xffffffffull;
        if (last) totals = tot >> 32;
    }
    unsigned long long ti_tot;
    block_excl_scan64((unsigned long long)true_inst, s_tmp, ti_tot);
    unsigned long long tiles_tot;                                // 16x16 tile instances (upstream's num_rendered)
    {
        unsigned long long tl = 0ull;
        for (int c = tid; c < chunks; c += SCAN_THREADS) tl += w.chunk_tiles[c];
        block_excl_scan64(tl, s_tmp, tiles_tot);
    }
    if (tid == 0) {
        w.cell_off[cells] = make_uint2((uint32_t)carry, (uint32_t)(carry >> 32));
        w.header->num_rendered = (uint32_t)(carry >> 32);
        w.header->overflow = 0u;
        w.header->max_tile_list = (uint32_t)carry;     // reused slot: number of (Gaussian, cell) entries
        w.header->num_visible = (uint32_t)totals;
        w.header->num_instances = (uint32_t)ti_tot;
        w.header->num_tile_instances = (uint32_t)tiles_tot;
    }
    for (int i = tid; i < XCD_REGIONS * ORDER_CLASSES; i += SCAN_THREADS) w.cls_cur[i] = 0u;
            if (tid < 4) w.bwd_meta[tid] = 0u;
    __syncthreads();                                             // cell_off (all of it) visible to the whole workgroup
    write_cell_order(w, cells, [&](int c) { return (uint32_t)((c == tid ? v0 : w.cell_cnt[c]) >> 32); });
}

// Exact footprint test.  The sub-tile rect of a splat (preprocess_fwd.hip) is the bounding box of {alpha >= 1/255}; an
// ellipse leaves the corners of its box empty, and for avatar-sized splats (a box of 2x3 sub-tiles) that is 20 % of all
// (splat, sub-tile) instances (C3 view 0: 893 k -> 709 k).  A sub-tile whose 8x8 pixel centres all fail the per-pixel
// alpha test contributes nothing to the image or to any gradient, so it never enters a list.  Per ROW of sub-tiles (a band
// of 8 pixel rows) the sub-tiles the ellipse reaches are those whose pixel-centre range meets its x-extent inside the band
// (common.h: make_footprint / footprint_band): O(1) per row instead of a test per sub-tile -- a large scene splat crosses up
// to 64 sub-tiles of a cell.
// CHUNK Gaussians per workgroup: (a) Gaussian-major instance offsets (in-chunk prefix + chunk_off) stored
// into the splat record, (b) 16-byte entries scattered into their cells' buckets: the chunk's first slot in
// every cell comes from the scanned count matrix, ranks inside it from LDS atomics, (c) clears this
// workgroup's slice of the batch-owner array.
constexpr int SC_BLOCK = CHUNK;        // one Gaussian per thread
// Zero-copy header report (ExaRasterForwardJob.host_header): ONE 16-byte write-through store (sc0 sc1 = system scope) into
// host-coherent pinned memory -- a single bus transaction, so a host that sees the tag sees the three values, without
// a release fence (which would write the whole L2 of this XCD back first).
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void report_header(uint32_t* host_hdr, uint32_t need, uint32_t overflow, uint32_t vis, uint32_t tag) {
    const v4u v = {need, overflow, vis, tag};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(host_hdr), "v"(v) : "memory");
}
__global__ __launch_bounds__(SC_BLOCK) void cell_scatter_kernel(Batch<BinArgs> batch) {
    // base[cells] | cnt2[cells] (u32)  [ | tot[cells] | bef[cells] (u64) when the scans are merged into this kernel ]
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    __shared__ uint32_t s_tmp[SC_BLOCK / 64];
    __shared__ unsigned long long s_tmp64[SC_BLOCK / 64];
    __shared__ uint32_t s_bcast[2];
    const BinArgs& a = batch.v[blockIdx.y];
    const int P = a.P;
    Splat* __restrict__ splats = a.splats;
    const TileWs& w = a.tw;
    const Grid& g = a.grid;
    const BinWs& b = a.bw;
    const uint64_t capacity = a.capacity;
    const int tid = threadIdx.x, cells = g.cells;
#ifdef EXA_PROBE_SCATTER   // probe build only (tools/gpu_scatter_phases.py): phases of every workgroup on the 100 MHz clock
    const unsigned long long ps_t0 = wall_clock64();
#define SCATTER_PHASE(i) do { __syncthreads(); if (tid == 0) w.part_cnt[40000 + 8 * blockIdx.x + (i)] = (uint32_t)(wall_clock64() - ps_t0); } while (0)
    if (tid == 0) { w.part_cnt[40000 + 8 * blockIdx.x + 6] = (uint32_t)ps_t0; }
#else
#define SCATTER_PHASE(i) do { } while (0)
#endif
    {   // this workgroup's slice of the zero-filled section of the bin workspace: batch owners (written by
        // subtile_bin_kernel, the next launch), blended masks (render_fwd), touched bytes (render_bwd)
        const size_t n16 = bin_zero_bytes(capacity) / 16, per = (n16 + gridDim.x - 1) / gridDim.x;
        const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
        for (size_t i = lo + threadIdx.x; i < hi; i += SC_BLOCK) b.owner[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    SCATTER_PHASE(0);                                            // zero-fill slice issued
    uint32_t* s_base = s_dyn;
    uint32_t* s_cnt2 = s_dyn + cells;
    // Merged scans: the workgroup AFTER the job's last chunk has no Gaussians to scatter; it is the one that publishes what
    // the later kernels read (cell_off, header, cell order), off the critical path of the scattering workgroups.
    const bool publisher = a.merged && (int)blockIdx.x == a.chunks;
    if ((int)blockIdx.x >= a.chunks && !publisher) {            // a job with fewer Gaussians than the largest of the batch
        if (!a.merged && a.chunks == 0 && blockIdx.x == 0 && tid == 0 && a.host_hdr)
            report_header(a.host_hdr, w.header->num_rendered, 0u, 0u, a.hdr_tag);     // (cell_scan wrote the header)
        return;
    }
    // This chunk's splat rows depend on nothing but the chunk: they are requested right after the column walk, so that the
    // trip overlaps the block scans (phase probe: 2.3 us between the scans and the scatter, 1.2 of them this trip; requested
    // BEFORE the walk they only delayed it -- loads return in order).
    constexpr int PER = CHUNK / SC_BLOCK;
    uint4 r3[PER];
    int ids[PER];
    uint32_t depth_bits[PER];
    uint32_t mine = 0;
    auto load_rows = [&] {
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            ids[it] = blockIdx.x * CHUNK + it * SC_BLOCK + tid;
            r3[it] = make_uint4(0, 0, 0, 0);
            depth_bits[it] = 0;
            if (ids[it] < P) {
                r3[it] = reinterpret_cast<const uint4*>(splats + ids[it])[3];
                depth_bits[it] = reinterpret_cast<const uint4*>(splats + ids[it])[2].w;
            }
        }
    };
    uint32_t D, chunk_off;
    if (a.merged) {
        // The scans of the (chunk, cell) count matrix, done redundantly by EVERY scatter workgroup instead of by two
        // tiny launches in front of it (col_scan + cell_scan: ~12 us of pure launch and memory latency for C3): column
        // totals, this chunk's entry offset in every cell, the prefix over the cells, the prefix of the chunks'
        // instance counts.  The matrix (chunks x cells x 8 B, <= 512 KB here) is L2-resident; workgroup 0 also publishes
        // what the later kernels read (cell_off, header, cell order).
        unsigned long long* s_tot = reinterpret_cast<unsigned long long*>(s_dyn + 2 * cells);
        unsigned long long* s_bef = s_tot + cells;
        // (requested before the column walk: between the two block scans this load was a dependent trip of ~1 us)
        const uint32_t ci = tid < a.chunks ? w.chunk_inst[tid] : 0u;
        for (int c = tid; c < cells; c += SC_BLOCK) { s_tot[c] = 0ull; s_bef[c] = 0ull; }
        __syncthreads();
        {
            if ((cells & 1) == 0) {                             // two cells per thread and row: 16-byte loads, twice the row groups
                const int cp = cells >> 1, G2 = SC_BLOCK / cp;   // (5.2 -> 4.0 us of this kernel's 13.9 on C3; four cells per thread: the
                                                                 //  same; ten rows in flight: 5.1; rows packed to u32 `inst << 11 | entries`: 5.5)
                const int c2 = tid % cp, q = tid / cp;
                if (q < G2) {
                    unsigned long long tot0 = 0ull, tot1 = 0ull, bef0 = 0ull, bef1 = 0ull;
                    const ulonglong2* m = reinterpret_cast<const ulonglong2*>(w.chunk_cell) + c2;
#pragma unroll 4
                    for (int r = q; r < a.chunks; r += G2) {
                        const ulonglong2 v = m[(size_t)r * cp];
                        const unsigned long long lowmask = r < (int)blockIdx.x ? 0xffffffffull : 0ull;
                        tot0 += v.x; tot1 += v.y;
                        bef0 += v.x & lowmask; bef1 += v.y & lowmask;
                    }
                    __hip_atomic_fetch_add(&s_tot[2 * c2], tot0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&s_tot[2 * c2 + 1], tot1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&s_bef[2 * c2], bef0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&s_bef[2 * c2 + 1], bef1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
            const int G = SC_BLOCK / cells;                     // row groups (cells <= SC_BLOCK on this path)
            const int c = tid % cells, q = tid / cells;
            if (q < G) {
                unsigned long long tot = 0ull, bef = 0ull;
                const unsigned long long* m = w.chunk_cell + c;
#pragma unroll 4
                for (int r = q; r < a.chunks; r += G) {
                    const unsigned long long v = m[(size_t)r * cells];
                    tot += v;
                    bef += r < (int)blockIdx.x ? (v & 0xffffffffull) : 0ull;
                }
                __hip_atomic_fetch_add(&s_tot[c], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&s_bef[c], bef, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            }
        }
        if (!publisher) load_rows();
        __syncthreads();
        SCATTER_PHASE(1);                                        // column walk of the count matrix
        const unsigned long long v = tid < cells ? s_tot[tid] : 0ull;
        const unsigned long long packed = ((unsigned long long)(cell_slots((uint32_t)(v >> 32)) * BATCH) << 32) | (uint32_t)v;
        // (slot space << 32 | entries) prefix over the cells, and the instances of the chunks in front of this one
        // (chunks <= SC_BLOCK on this path): both scans behind one pair of barriers
        unsigned long long tot_all, x;
        uint32_t inst_total, cx;
        block_excl_scan_pair(packed, ci, s_tmp64, s_tmp, x, tot_all, cx, inst_total);
        if (tid < cells) s_base[tid] = (uint32_t)x + (uint32_t)s_bef[tid];
        D = (uint32_t)(tot_all >> 32);
        if (tid == (int)blockIdx.x) s_bcast[0] = cx;
        if (publisher) {
            if (tid < cells) w.cell_off[tid] = make_uint2((uint32_t)x, (uint32_t)(x >> 32));
            unsigned long long vt;                               // tiles << 32 | visible (both sums stay < 2^32)
            block_excl_scan64(tid < a.chunks ? ((unsigned long long)w.chunk_tiles[tid] << 32) | w.chunk_vis[tid] : 0ull, s_tmp64, vt);
            const uint32_t vis_total = (uint32_t)vt;
            if (tid == 0) {
                w.cell_off[cells] = make_uint2((uint32_t)tot_all, (uint32_t)(tot_all >> 32));
                w.header->num_rendered = D;
                w.header->overflow = (uint64_t)D > capacity ? 1u : 0u;
                w.header->max_tile_list = (uint32_t)tot_all;     // reused slot: number of (Gaussian, cell) entries
                w.header->num_visible = vis_total;
                w.header->num_instances = inst_total;
                w.header->num_tile_instances = (uint32_t)(vt >> 32);
                if (a.host_hdr) report_header(a.host_hdr, D, (uint64_t)D > capacity ? 1u : 0u, vis_total, a.hdr_tag);
            }
            for (int i = tid; i < XCD_REGIONS * ORDER_CLASSES; i += SC_BLOCK) w.cls_cur[i] = 0u;
            if (tid < 4) w.bwd_meta[tid] = 0u;
            __syncthreads();                                     // cell_off (all of it) visible to the whole workgroup
            write_cell_order(w, cells, [&](int c) { return (uint32_t)(s_tot[c] >> 32); });
            SCATTER_PHASE(5);                                    // publisher: everything
            return;
        }
        __syncthreads();
        chunk_off = s_bcast[0];
        SCATTER_PHASE(2);                                        // block scans
        if ((uint64_t)D > capacity) return;                      // overflow latched in the header by the publisher
        for (int c = tid; c < cells; c += SC_BLOCK) s_cnt2[c] = 0u;
    } else {
        load_rows();
        D = w.header->num_rendered;
        if (a.host_hdr && blockIdx.x == 0 && threadIdx.x == 0)
            report_header(a.host_hdr, D, (uint64_t)D > capacity ? 1u : 0u, w.header->num_visible, a.hdr_tag);
        if ((uint64_t)D > capacity) {
            if (blockIdx.x == 0 && threadIdx.x == 0) w.header->overflow = 1u;
            return;
        }
        chunk_off = w.chunk_off[blockIdx.x];
        for (int c = tid; c < cells; c += SC_BLOCK) {
            s_cnt2[c] = 0u;
            s_base[c] = w.cell_off[c].x + (uint32_t)w.chunk_cell[(size_t)blockIdx.x * cells + c];
        }
    }
    __syncthreads();

#pragma unroll
    for (int it = 0; it < PER; ++it) mine += r3[it].z;
    uint32_t total;
    uint32_t off = chunk_off + block_excl_scan(mine, s_tmp, total);
    SCATTER_PHASE(3);                                            // splat rows loaded, in-chunk prefix
#pragma unroll
    for (int it = 0; it < PER; ++it) {
        if (r3[it].z) {
            splats[ids[it]].inst_off = off;
            off += r3[it].z;
            const int sx0 = r3[it].x & 0xffff, sx1 = r3[it].x >> 16, sy0 = r3[it].y & 0xffff, sy1 = r3[it].y >> 16;
            for (int cy = sy0 >> 3; cy <= (sy1 - 1) >> 3; ++cy)
                for (int cx = sx0 >> 3; cx <= (sx1 - 1) >> 3; ++cx) {
                    const int c = cy * g.cx + cx;
                    const uint32_t r = __hip_atomic_fetch_add(&s_cnt2[c], 1u, __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_WORKGROUP);
                    b.bucket[s_base[c] + r] = make_uint4((uint32_t)ids[it], depth_bits[it], r3[it].x, r3[it].y);
                }
        }
    }
    SCATTER_PHASE(4);                                            // entries scattered
}

// One 1024-thread workgroup per cell: second radix digit.  Counts the cell's entries per 8x8 sub-tile in
// LDS, turns the counts into [begin, end) ranges (cell-major sub-tile order) and scatters the sort keys.
// Entries are 16-byte records read coalesced; no gather from the splat array.
constexpr int BIN_THREADS = 1024;
// (SINGLE_PART_CELLS = 1024, common.h: from this many cells on (2048 x 2048 px) one workgroup per cell)
// An avatar's instances sit in a few dozen cells, and both halves of this digit -- LDS counting atomics and the
// scattered 8-byte key stores, one per instance -- are throughput limits of ONE CU.  Every cell is therefore
// handled by BIN_PARTS workgroups in two launches:
//   subtile_count_kernel  part p counts its quarter of the cell's entries per sub-tile -> part_cnt[cell][p][64]
//   subtile_bin_kernel    reads the cell's BIN_PARTS x 64 counts (totals -> 64-aligned ranges; earlier parts ->
//                         its own first slot in every sub-tile) and scatters its quarter of the keys.
// Plain stores and a kernel boundary instead of any cross-workgroup atomics; part 0 publishes ranges and owners.
// (Round 3 tried ONE launch: footprint masks computed by cell_scatter_kernel -- the record is in registers there -- and
//  every part counting the whole cell itself.  Bitwise the same lists, but slower: the mask work is Gaussian-major there,
//  one thread walks ALL cells of its splat -- C3 cell_scatter 18.4 -> 22.8 us for 19.7 -> 16.8 us here, C5 (scene splats
//  over hundreds of cells) 20.7 -> 76 us.  Entry-parallel masks in their own launch stay.)
// (Round 3 also tried to fuse the scatter below with the SORT of the lists (commit 10d6018, removed again): one 512-thread
//  workgroup per row of eight sub-tiles scans the cell's entries, keeps the keys of its row in LDS and every wave sorts one
//  list.  The keys never touch HBM and one launch goes, but it is slower -- count 11.6 + fused 30 us against bin 19.6 +
//  sort 18.3 us: the dense rows make their CUs instruction-issue bound (eight-fold scan, eight lists per CU) while this
//  version spreads the same work over the chip.  Keeping several entries in flight per thread changes nothing in any of
//  these loops either: they are not load-latency bound.)
struct CellPart { int cell, part; uint32_t e0, e1, lo, hi, slot0; bool overflow, active; int rank, nparts; };
// the work record of this workgroup from the table write_part_table left (ONE load, next to the header's); false: no work
__device__ __forceinline__ bool cell_part_of(const TileWs& w, uint64_t capacity, CellPart& c) {
    const uint4 d = w.part_desc[blockIdx.x];
    const uint32_t need = w.header->num_rendered;
    if (d.x == NO_PART) return false;
    c.cell = (int)(d.x & 0xfffu); c.rank = (int)((d.x >> 12) & 0xfffu);
    c.part = (int)((d.x >> 24) & 0xfu); c.nparts = (int)(d.x >> 28) + 1;
    c.overflow = (uint64_t)need > capacity;
    c.lo = d.y; c.hi = c.overflow ? d.y : d.z;
    c.e0 = c.e1 = 0u;                                           // (only the one-workgroup-per-cell kernel uses them)
    c.slot0 = d.w;
    c.active = d.z > d.y || c.nparts > 1 || c.part > 0;          // a cell without entries has ONE part with an empty share
    return true;
}
template <int PARTS>
__device__ __forceinline__ CellPart cell_part(const TileWs& w, uint64_t capacity) {   // blockIdx.x < cells * PARTS
    CellPart c;
    // the header and the cell record are independent loads: one round trip
    const uint4 d = w.cell_desc[blockIdx.x / PARTS];
    const uint32_t active = w.header->active_cells, need = w.header->num_rendered;
    c.cell = (int)d.x;
    c.part = (int)(blockIdx.x % PARTS);
    c.active = blockIdx.x / PARTS < active;
    c.rank = (int)(blockIdx.x / PARTS); c.nparts = PARTS;
    c.overflow = (uint64_t)need > capacity;
    c.e0 = d.y;
    c.e1 = c.overflow ? d.y : d.z;
    c.slot0 = d.w;
    const uint32_t per = (c.e1 - c.e0 + PARTS - 1) / PARTS;
    c.lo = min(c.e1, c.e0 + (uint32_t)c.part * per);
    c.hi = min(c.e1, c.lo + per);
    return c;
}

// The sub-tiles of its cell (origin csx0, csy0 in sub-tiles) that the footprint of entry `en` = {id, depth, rect x, rect y}
// really reaches, as a 64-bit mask (bit = y * 8 + x).  It replaces the rect in the entry; the scatter walks the same bits.
template <bool FOOTPRINT>
__device__ __forceinline__ unsigned long long entry_mask(const Splat* __restrict__ splats, const uint4& en, int csx0, int csy0) {
    const uint4* rec = reinterpret_cast<const uint4*>(splats + en.x);
    uint4 r0 = make_uint4(0u, 0u, 0u, 0u), r1 = r0;              // (A = 0: no test, the whole rect)
    if (FOOTPRINT) { r0 = rec[0]; r1 = rec[1]; }
    const int x0 = max((int)(en.z & 0xffff) - csx0, 0), x1 = min((int)(en.z >> 16) - csx0, CELL_SUBS);
    const int y0 = max((int)(en.w & 0xffff) - csy0, 0), y1 = min((int)(en.w >> 16) - csy0, CELL_SUBS);
    const float xl0 = (float)((csx0 + x0) * SUB) - __uint_as_float(r0.x), yl0 = (float)((csy0 + y0) * SUB) - __uint_as_float(r0.y);
    const Footprint fp = make_footprint(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z), __uint_as_float(r1.w),
                                        fmaxf(fabsf(xl0), fabsf(xl0 + (float)((x1 - x0) * SUB))),
                                        fmaxf(fabsf(yl0), fabsf(yl0 + (float)((y1 - y0) * SUB))));
    unsigned long long mask = 0ull;
    for (int y = y0; y < y1; ++y) {
        int c0 = x0, c1 = x1 - 1;
        if (fp.test) {
            float xa, xb;
            const float yl = yl0 + (float)((y - y0) * SUB);
            if (!footprint_band(fp, yl, yl + (float)(SUB - 1), xa, xb)) continue;
            if (xa <= xb) {                                     // (NaN: keep the row)
                const float m = 1e-3f * (1.0f + fmaxf(fabsf(xa), fabsf(xb)));
                const float lo = clampf((xa - m - (float)(SUB - 1) - xl0) * (1.0f / SUB), -1.0e6f, 1.0e6f);
                const float hi = clampf((xb + m - xl0) * (1.0f / SUB), -1.0e6f, 1.0e6f);
                c0 = max(x0, x0 + (int)ceilf(lo));
                c1 = min(x1 - 1, x0 + (int)floorf(hi));
            }
        }
        if (c1 < c0) continue;
        mask |= (unsigned long long)((2u << c1) - (1u << c0)) << (y * CELL_SUBS);
    }
    return mask;
}

template <bool FOOTPRINT, int PARTS>
__global__ __launch_bounds__(BIN_THREADS) void subtile_count_kernel(Batch<BinArgs> batch) {
    __shared__ uint32_t s_cnt[SUBS_PER_CELL];
    const BinArgs& a = batch.v[blockIdx.y];
    const TileWs& w = a.tw;
    const Grid& g = a.grid;
    const BinWs& b = a.bw;
    if ((int)blockIdx.x >= g.cells * PARTS) return;
    CellPart cp;
    if (!cell_part_of(w, a.capacity, cp) || !cp.active) return;         // no work / empty cell: nothing to count
    const int tid = threadIdx.x;
    if (tid < SUBS_PER_CELL) s_cnt[tid] = 0u;
    __syncthreads();
    const int csx0 = (cp.cell % g.cx) * CELL_SUBS, csy0 = (cp.cell / g.cx) * CELL_SUBS;   // cell origin in sub-tiles
    for (uint32_t e = cp.lo + tid; e < cp.hi; e += BIN_THREADS) {
        // ONE 16-byte load of the entry: left to itself the compiler fetched the rect words first and sank the id word
        // into the block that uses it -- behind a wait: a fourth dependent round trip in a launch that consists of four
        uint4 en = b.bucket[e];
        asm("" : "+v"(en.x) : "v"(en.y), "v"(en.z), "v"(en.w));
        const unsigned long long mask = entry_mask<FOOTPRINT>(a.splats, en, csx0, csy0);
        for (unsigned long long m = mask; m; m &= m - 1)
            __hip_atomic_fetch_add(&s_cnt[__builtin_ctzll(m)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        reinterpret_cast<uint2*>(b.bucket + e)[1] = make_uint2((uint32_t)mask, (uint32_t)(mask >> 32));
    }
    __syncthreads();
    if (tid < SUBS_PER_CELL) w.part_cnt[(size_t)blockIdx.x * SUBS_PER_CELL + tid] = s_cnt[tid];       // by workgroup
}

template <int PARTS>
__global__ __launch_bounds__(BIN_THREADS) void subtile_bin_kernel(Batch<BinArgs> batch) {
    __shared__ uint32_t s_off[SUBS_PER_CELL];
    __shared__ uint32_t s_cnt2[SUBS_PER_CELL];
    const BinArgs& a = batch.v[blockIdx.y];
    const TileWs& w = a.tw;
    const Grid& g = a.grid;
    const BinWs& b = a.bw;
    if ((int)blockIdx.x >= g.cells * PARTS) return;
    CellPart cp;
    if (!cell_part_of(w, a.capacity, cp)) return;
    // empty cell: its one workgroup publishes 64 empty ranges.  So does part 0 of every cell of an OVERFLOWED render, and
    // nothing else of it runs: cell_scatter left the buckets unwritten, and the cell's entry offsets (cp.lo) count entries of
    // the render that did not fit -- they lie beyond the `capacity` entries the bucket array holds.  Until round 5 the
    // prefetch below was issued all the same, `bucket[cp.lo - 1]` = up to (entries - capacity) * 16 bytes past the end of the
    // bin workspace: harmless while the allocator has something mapped behind it, a GPU memory fault (SIGABRT of the
    // process at its next runtime call) when it has not -- the round-4 GPUTEST abort.
    if (!cp.active || cp.overflow) {
        if (cp.part == 0) {
            if (threadIdx.x < SUBS_PER_CELL) {
                w.ranges[cp.cell * SUBS_PER_CELL + threadIdx.x] = make_uint2(0u, 0u);
                w.cls_code[cp.cell * SUBS_PER_CELL + threadIdx.x] = (uint8_t)0;
            }
            if (threadIdx.x == 0) w.cell_long[cp.rank] = 0u;
        }
        return;
    }
    const int cell = cp.cell, tid = threadIdx.x;
    // this thread's first entry is requested NOW: it depends on the part record only, and behind the barrier below it
    // was a round trip of its own (the launch consists of three)
    const uint32_t e_first = cp.lo + (uint32_t)tid;
    // (unconditional, clamped into the cell's entries: a conditional load would be waited for at the join right here)
    const uint4 en_first = b.bucket[min(e_first, max(cp.hi, 1u) - 1u)];
    if (tid < 64) {
        uint32_t n = 0, before = 0;
        const size_t first = (size_t)blockIdx.x - (size_t)cp.part;      // the cell's parts are consecutive workgroups
#pragma unroll 4
        for (int p = 0; p < cp.nparts; ++p) {
            const uint32_t v = w.part_cnt[(first + p) * SUBS_PER_CELL + tid];
            before += p < cp.part ? v : 0u;
            n += v;
        }
        const uint32_t nslot = n ? (n + BATCH - 1) / BATCH + 1 : 0u;       // real batches + one end slot
        const uint32_t incl = wave_incl_scan(nslot);
        const uint32_t begin = cp.overflow ? 0u : cp.slot0 + (incl - nslot) * BATCH;
        s_off[tid] = begin + before;
        s_cnt2[tid] = 0u;
        if (cp.part == 0) {
            const unsigned long long longer = __ballot(n > (uint32_t)BATCH);
            if (tid == 0) w.cell_long[cp.rank] = (uint32_t)__popcll(longer);
            w.ranges[cell * SUBS_PER_CELL + tid] = make_uint2(begin, begin + n);
            w.cls_code[cell * SUBS_PER_CELL + tid] = (uint8_t)length_class(n);
            for (uint32_t bq = 0; bq + 1 < nslot; ++bq)
                b.owner[begin / BATCH + bq] = make_uint4((uint32_t)(cell * SUBS_PER_CELL + tid) + 1u, begin, n, 0u);
        }
    }
    __syncthreads();
    auto scatter = [&](const uint4& en) {
        const unsigned long long key = ((unsigned long long)en.y << 32) | en.x;
        for (unsigned long long m = ((unsigned long long)en.w << 32) | en.z; m; m &= m - 1) {
            const int s = __builtin_ctzll(m);
            const uint32_t r = __hip_atomic_fetch_add(&s_cnt2[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            b.keys[s_off[s] + r] = key;
        }
    };
    if (e_first < cp.hi) scatter(en_first);
    for (uint32_t e = e_first + BIN_THREADS; e < cp.hi; e += BIN_THREADS) scatter(b.bucket[e]);
}

// One workgroup per cell, both halves in one launch: for images with >= SINGLE_PART_CELLS cells (2048 x 2048 px) there are
// enough cells to fill the chip without splitting them, and the split only multiplies the fixed costs.
template <bool FOOTPRINT>
__global__ __launch_bounds__(BIN_THREADS) void subtile_count_bin_kernel(Batch<BinArgs> batch) {
    __shared__ uint32_t s_cnt[SUBS_PER_CELL], s_off[SUBS_PER_CELL], s_cnt2[SUBS_PER_CELL];
    const BinArgs& a = batch.v[blockIdx.y];
    const TileWs& w = a.tw;
    const Grid& g = a.grid;
    const BinWs& b = a.bw;
    if ((int)blockIdx.x >= g.cells) return;
    const CellPart cp = cell_part<1>(w, a.capacity);
    const int cell = cp.cell, tid = threadIdx.x;
    if (!cp.active) {                                                   // empty cell: 64 empty ranges
        if (tid < SUBS_PER_CELL) {
            w.ranges[cell * SUBS_PER_CELL + tid] = make_uint2(0u, 0u);
            w.cls_code[cell * SUBS_PER_CELL + tid] = (uint8_t)0;
        }
        if (tid == 0) w.cell_long[blockIdx.x] = 0u;
        return;
    }
    if (tid < SUBS_PER_CELL) s_cnt[tid] = 0u;
    __syncthreads();
    const int csx0 = (cell % g.cx) * CELL_SUBS, csy0 = (cell / g.cx) * CELL_SUBS;   // cell origin in sub-tiles
    for (uint32_t e = cp.lo + tid; e < cp.hi; e += BIN_THREADS) {
        // ONE 16-byte load of the entry: left to itself the compiler fetched the rect words first and sank the id word
        // into the block that uses it -- behind a wait: a fourth dependent round trip in a launch that consists of four
        uint4 en = b.bucket[e];
        asm("" : "+v"(en.x) : "v"(en.y), "v"(en.z), "v"(en.w));
        const unsigned long long mask = entry_mask<FOOTPRINT>(a.splats, en, csx0, csy0);
        for (unsigned long long m = mask; m; m &= m - 1)
            __hip_atomic_fetch_add(&s_cnt[__builtin_ctzll(m)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        reinterpret_cast<uint2*>(b.bucket + e)[1] = make_uint2((uint32_t)mask, (uint32_t)(mask >> 32));   // read back by this thread
    }
    __syncthreads();
    if (tid < 64) {
        const uint32_t n = s_cnt[tid];
        const uint32_t nslot = n ? (n + BATCH - 1) / BATCH + 1 : 0u;       // real batches + one end slot
        const uint32_t incl = wave_incl_scan(nslot);
        const uint32_t begin = cp.overflow ? 0u : cp.slot0 + (incl - nslot) * BATCH;
        s_off[tid] = begin;
        s_cnt2[tid] = 0u;
        const unsigned long long longer = __ballot(n > (uint32_t)BATCH);
        if (tid == 0) w.cell_long[blockIdx.x] = (uint32_t)__popcll(longer);
        w.ranges[cell * SUBS_PER_CELL + tid] = make_uint2(begin, begin + n);
        w.cls_code[cell * SUBS_PER_CELL + tid] = (uint8_t)length_class(n);
        for (uint32_t bq = 0; bq + 1 < nslot; ++bq)
            b.owner[begin / BATCH + bq] = make_uint4((uint32_t)(cell * SUBS_PER_CELL + tid) + 1u, begin, n, 0u);
    }
    __syncthreads();
    for (uint32_t e = cp.lo + tid; e < cp.hi; e += BIN_THREADS) {
        const uint4 en = b.bucket[e];
        const unsigned long long key = ((unsigned long long)en.y << 32) | en.x;
        for (unsigned long long m = ((unsigned long long)en.w << 32) | en.z; m; m &= m - 1) {
            const int s = __builtin_ctzll(m);
            const uint32_t r = __hip_atomic_fetch_add(&s_cnt2[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            b.keys[s_off[s] + r] = key;
        }
    }
}

hipError_t launch_zero(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    const size_t n16 = (bytes + 15) / 16;        // workspaces are 256-byte padded, rounding up is safe
    const int blocks = (int)((n16 + BLOCK - 1) / BLOCK < 2048 ? (n16 + BLOCK - 1) / BLOCK : 2048);
    zero_kernel<<<blocks, BLOCK, 0, s>>>(static_cast<uint4*>(p), n16);
    return hipGetLastError();
}

hipError_t launch_cell_scan(const BinArgs* a, int K, hipStream_t s) {
    const Batch<BinArgs> b = make_batch(a, K);
    int cells = 0;
    for (int k = 0; k < K; ++k) cells = max(cells, a[k].grid.cells);
    if (cells > 0) col_scan_kernel<<<dim3((cells + COL_W - 1) / COL_W, K), SCAN_THREADS, 0, s>>>(b);
    cell_scan_kernel<<<dim3(1, K), SCAN_THREADS, 0, s>>>(b);
    return hipGetLastError();
}

hipError_t launch_cell_scatter(const BinArgs* a, int K, hipStream_t s) {
    const Batch<BinArgs> b = make_batch(a, K);
    int chunks = 0, cells = 0;
    for (int k = 0; k < K; ++k) {
        chunks = max(chunks, a[k].chunks);
        cells = max(cells, a[k].grid.cells);
    }
    // one workgroup past the last chunk: the publisher of the merged scans; without them at least one workgroup per job,
    // which clears the zero-filled section
    chunks = a[0].merged ? chunks + 1 : max(chunks, 1);
    cell_scatter_kernel<<<dim3(chunks, K), SC_BLOCK, (size_t)cells * (a[0].merged ? 24 : 8), s>>>(b);
    return hipGetLastError();
}

hipError_t launch_subtile_bin(const BinArgs* a, int K, hipStream_t s) {
    const Batch<BinArgs> b = make_batch(a, K);
    int cells = 0;
    for (int k = 0; k < K; ++k) cells = max(cells, a[k].grid.cells);
    if (cells == 0) return hipSuccess;
    const bool footprint = dev_knobs().footprint;
    // Workgroups per cell: an avatar view fills a few dozen of its 256 cells, so every cell is split over BIN_PARTS
    // workgroups (two launches) to get the chip busy; a large image (C5: 1024 cells, content everywhere) has enough
    // cells already and takes one workgroup per cell that counts and scatters in ONE launch.
    const int single_cells = dev_knobs().single_cells;
    if (cells >= single_cells || cells > 1024) {                  // (the part table maps one thread to one cell)
        if (footprint) subtile_count_bin_kernel<true><<<dim3(cells, K), BIN_THREADS, 0, s>>>(b);
        else subtile_count_bin_kernel<false><<<dim3(cells, K), BIN_THREADS, 0, s>>>(b);
    } else {
        if (footprint) subtile_count_kernel<true, BIN_PARTS><<<dim3(cells * BIN_PARTS, K), BIN_THREADS, 0, s>>>(b);
        else subtile_count_kernel<false, BIN_PARTS><<<dim3(cells * BIN_PARTS, K), BIN_THREADS, 0, s>>>(b);
        subtile_bin_kernel<BIN_PARTS><<<dim3(cells * BIN_PARTS, K), BIN_THREADS, 0, s>>>(b);
    }
    return hipGetLastError();
}

}  // namespace exa
