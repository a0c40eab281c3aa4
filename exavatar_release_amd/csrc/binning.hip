// Tile binning for gfx950: the "duplicate then radix sort" of the upstream rasterizer restructured
// as an MSD radix sort whose first digit is the tile id:
//   count  (in preprocess_fwd.hip)  per-(sub-counter, tile) instance counts           -- histogram
//   scan   (tile_scan_kernel)       exclusive prefix over (tile, sub) -> bucket cursors and ranges
//   scatter(scatter_kernel)         every Gaussian writes (depth bits << 32 | id) into its tiles' buckets
// The remaining 32 depth bits are sorted per bucket inside LDS by the render kernel
// (render_fwd.hip), which reproduces the order of upstream's stable global sort on
// (tile << 32 | depth bits): ascending depth, ties by ascending Gaussian id.
//
// Replaces upstream InclusiveSum + duplicateWithKeys + SortPairs(tile digit) + identifyTileRanges
// (SURVEY.md section 2.1).  HBM traffic: scan 2 * 4 * NSUB * tiles B; scatter reads 32 B of each
// visible splat record and writes 8 B per instance.
#include "common.h"

namespace exa {

constexpr int SCAN_THREADS = 1024;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// One workgroup: tiles*NSUB counters -> cursors (exclusive prefix in (tile, sub) order), tile ranges,
// header {num_rendered, max_tile_list}.
__global__ __launch_bounds__(SCAN_THREADS) void tile_scan_kernel(TileWs w, int tiles) {
    __shared__ uint32_t s_wave[SCAN_THREADS / 64];
    __shared__ uint32_t s_max[SCAN_THREADS / 64];
    const int tid = threadIdx.x;
    const int per = (tiles + SCAN_THREADS - 1) / SCAN_THREADS;
    const int t0 = min(tid * per, tiles), t1 = min(t0 + per, tiles);
    uint32_t sum = 0, mx = 0;
    for (int t = t0; t < t1; ++t) {
        uint32_t ts = 0;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) ts += w.counts[(size_t)s * tiles + t];
        sum += ts;
        mx = max(mx, ts);
    }
    const uint32_t incl = wave_incl_scan(sum);
    uint32_t wmx = mx;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wmx = max(wmx, __shfl_xor(wmx, d, 64));
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 63) { s_wave[wave] = incl; s_max[wave] = wmx; }
    __syncthreads();
    uint32_t base = 0, total = 0, gmx = 0;
#pragma unroll
    for (int i = 0; i < SCAN_THREADS / 64; ++i) {
        const uint32_t ws = s_wave[i];
        if (i < wave) base += ws;
        total += ws;
        gmx = max(gmx, s_max[i]);
    }
    uint32_t run = base + incl - sum;
    for (int t = t0; t < t1; ++t) {
        const uint32_t begin = run;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const size_t o = (size_t)s * tiles + t;
            w.cursor[o] = run;
            run += w.counts[o];
        }
        w.ranges[t] = make_uint2(begin, run);
    }
    if (tid == 0) {
        w.header->num_rendered = total;
        w.header->max_tile_list = gmx;
    }
}

// One thread per Gaussian: emit (depth bits << 32 | id) into every touched tile's bucket.
__global__ __launch_bounds__(BLOCK) void scatter_kernel(int P, const Splat* __restrict__ splats, TileWs w, int tiles,
                                                        int gx, BinWs b, uint64_t capacity) {
    const uint32_t D = w.header->num_rendered;
    if ((uint64_t)D > capacity) {
        if (blockIdx.x == 0 && threadIdx.x == 0) w.header->overflow = 1u;
        return;
    }
    const int idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= P) return;
    const uint4* rec = reinterpret_cast<const uint4*>(splats + idx);
    const uint4 r3 = rec[3];
    if (r3.z == 0) return;
    const uint4 r0 = rec[0];
    const unsigned long long key = ((unsigned long long)r0.z << 32) | (uint32_t)idx;
    const int x0 = r3.x & 0xffff, x1 = r3.x >> 16, y0 = r3.y & 0xffff, y1 = r3.y >> 16;
    uint32_t* cur = w.cursor + (size_t)sub_of(idx) * tiles;
    for (int ty = y0; ty < y1; ++ty)
        for (int tx = x0; tx < x1; ++tx) {
            const uint32_t pos = __hip_atomic_fetch_add(cur + ty * gx + tx, 1u, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
            b.keys[pos] = key;
        }
}

// Zero `n16` 16-byte words.  Used instead of hipMemsetAsync so that a captured hipGraph contains only
// kernel nodes (memset nodes of the bundled ROCm runtime misbehaved under capture / replay).
__global__ __launch_bounds__(BLOCK) void zero_kernel(uint4* __restrict__ p, size_t n16) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0, 0, 0, 0);
}

hipError_t launch_zero(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    const size_t n16 = (bytes + 15) / 16;        // workspaces are 256-byte padded, rounding up is safe
    const int blocks = (int)((n16 + BLOCK - 1) / BLOCK < 2048 ? (n16 + BLOCK - 1) / BLOCK : 2048);
    zero_kernel<<<blocks, BLOCK, 0, s>>>(static_cast<uint4*>(p), n16);
    return hipGetLastError();
}

hipError_t launch_tile_scan(const TileWs& w, int tiles, hipStream_t s) {
    tile_scan_kernel<<<1, SCAN_THREADS, 0, s>>>(w, tiles);
    return hipGetLastError();
}

hipError_t launch_scatter(int P, const Splat* splats, const TileWs& w, int tiles, int gx, const BinWs& b,
                          uint64_t capacity, hipStream_t s) {
    if (P == 0) return hipSuccess;
    scatter_kernel<<<(P + BLOCK - 1) / BLOCK, BLOCK, 0, s>>>(P, splats, w, tiles, gx, b, capacity);
    return hipGetLastError();
}

}  // namespace exa
