// Per-tile depth sort + front-to-back alpha compositing for gfx950.
//
// One 256-thread workgroup (4 waves) per 16x16 tile; wave w owns pixel rows 4w..4w+3 so every image
// row segment it touches is one 64-byte line.  Phase A sorts the tile's bucket of
// (depth bits << 32 | id) keys in LDS (bitonic, ascending-only network so a list of any length needs
// no padding) and, for training, writes the sorted ids once for the backward pass.  Phase B streams
// the sorted list in batches of 256: each thread gathers ONE 64-byte splat record (48 B used) into
// LDS, then all 256 pixels blend the batch from LDS broadcasts.
//
// Replaces the per-tile part of upstream SortPairs + renderCUDA (forward) of the rasterizer the
// reference calls at avatar/common/nets/module.py:632-640; per-pixel rule = oracle step 9/10
// (oracle/raster_oracle.py).
//
// Algorithmic HBM bytes: reads 8 B/instance (keys) + 48 B per gathered splat per tile it touches
// (L2-resident after the first touch), writes 4 B/instance (sorted ids, training only) and
// 20 B/pixel (rgb, depth, alpha) + 8 B/pixel (final_T, n_contrib, training only).
#include "common.h"

namespace exa {

template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort_asc(KeyPtr k, int n, int tid) {
    int m = 1;
    while (m < n) m <<= 1;
    const int pairs = m >> 1;
    for (int kk = 2; kk <= m; kk <<= 1) {
        const int half = kk >> 1;
        // flip stage: compare i with its mirror inside each kk-block (both halves ascending)
        for (int i = tid; i < pairs; i += BLOCK) {
            const int off = i & (half - 1);
            const int blk = (i - off) << 1;          // (i / half) * kk
            const int lo = blk + off, hi = blk + kk - 1 - off;
            if (hi < n) {
                const unsigned long long a = k[lo], b = k[hi];
                if (a > b) { k[lo] = b; k[hi] = a; }
            }
        }
        __syncthreads();
        for (int j = half >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < pairs; i += BLOCK) {
                const int lo = 2 * i - (i & (j - 1)), hi = lo + j;
                if (hi < n) {
                    const unsigned long long a = k[lo], b = k[hi];
                    if (a > b) { k[lo] = b; k[hi] = a; }
                }
            }
            __syncthreads();
        }
    }
}

template <bool STORE>
__global__ __launch_bounds__(BLOCK) void render_fwd_kernel(RenderFwdArgs a) {
    __shared__ unsigned long long s_keys[SORT_CAP];
    __shared__ float4 s_g0[BLOCK];
    __shared__ float4 s_g1[BLOCK];
    __shared__ float4 s_g2[BLOCK];
    __shared__ int s_done[BLOCK / 64];
    __shared__ uint32_t s_last[BLOCK / 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int tile_x = tile % a.grid.gx, tile_y = tile / a.grid.gx;
    const int pxi = tile_x * TILE + (tid & 15), pyi = tile_y * TILE + (tid >> 4);
    const bool inside = pxi < a.grid.W && pyi < a.grid.H;
    const float fx = (float)pxi, fy = (float)pyi;

    const bool overflow = (uint64_t)a.tw.header->num_rendered > a.capacity;
    if (overflow && blockIdx.x == 0 && tid == 0) a.tw.header->overflow = 1u;
    const uint2 range = a.tw.ranges[tile];
    const int n = overflow ? 0 : (int)(range.y - range.x);

    // ---- phase A: depth sort of this tile's bucket --------------------------------------------
    unsigned long long* gkeys = a.bw.keys + range.x;
    const bool in_lds = n <= SORT_CAP;
    if (n > 0) {
        if (in_lds) {
            for (int i = tid; i < n; i += BLOCK) s_keys[i] = gkeys[i];
            __syncthreads();
            if (n > 1) bitonic_sort_asc(s_keys, n, tid);
            if (STORE)
                for (int i = tid; i < n; i += BLOCK) a.bw.sorted[range.x + i] = (uint32_t)s_keys[i];
        } else {
            // rare: list longer than the LDS budget -> same network on the bucket in global memory
            bitonic_sort_asc(gkeys, n, tid);
            if (STORE)
                for (int i = tid; i < n; i += BLOCK) a.bw.sorted[range.x + i] = (uint32_t)gkeys[i];
        }
    }

    // ---- phase B: front-to-back blend ----------------------------------------------------------
    float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Dp = 0.f;
    uint32_t last = 0;
    bool done = !inside;
    const Splat* __restrict__ splats = a.splats;
    for (int base = 0; base < n; base += BLOCK) {
        const int wd = __all(done);
        if (lane == 0) s_done[wave] = wd;
        const int j = base + tid;
        if (j < n) {
            const uint32_t id = in_lds ? (uint32_t)s_keys[j] : (uint32_t)gkeys[j];
            const float4* rec = reinterpret_cast<const float4*>(splats + id);
            s_g0[tid] = rec[0];
            s_g1[tid] = rec[1];
            s_g2[tid] = rec[2];
        }
        __syncthreads();
        if (s_done[0] & s_done[1] & s_done[2] & s_done[3]) break;
        const int cnt = min(BLOCK, n - base);
        if (!wd) {
            for (int k = 0; k < cnt; ++k) {
                if (__all(done)) break;
                if (!done) {
                    const float4 g0 = s_g0[k];
                    const float4 g1 = s_g1[k];
                    const float dx = g0.x - fx, dy = g0.y - fy;
                    const float power = gauss_power(g1.x, g1.y, g1.z, dx, dy);
                    if (power <= 0.0f) {
                        const float alpha = fminf(ALPHA_MAX, g1.w * gauss_falloff(power));
                        if (alpha >= ALPHA_MIN) {
                            const float test_T = T * (1.0f - alpha);
                            if (test_T < T_EPS) {
                                done = true;
                            } else {
                                const float4 g2 = s_g2[k];
                                const float wgt = alpha * T;
                                Cr += g2.x * wgt; Cg += g2.y * wgt; Cb += g2.z * wgt;
                                Dp += g0.z * wgt;
                                T = test_T;
                                last = (uint32_t)(base + k + 1);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- outputs ---------------------------------------------------------------------------------
    const size_t HW = (size_t)a.grid.W * a.grid.H;
    if (inside) {
        const size_t pix = (size_t)pyi * a.grid.W + pxi;
        const float* __restrict__ bg = a.bg;
        a.out_color[pix] = Cr + T * bg[0];
        a.out_color[HW + pix] = Cg + T * bg[1];
        a.out_color[2 * HW + pix] = Cb + T * bg[2];
        a.out_depth[pix] = Dp;
        a.out_alpha[pix] = 1.0f - T;
        if (STORE) {
            a.iw.final_T[pix] = T;
            a.iw.n_contrib[pix] = last;
        }
    }
    if (STORE) {
        uint32_t wl = last;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) wl = max(wl, (uint32_t)__shfl_xor((int)wl, d, 64));
        __syncthreads();
        if (lane == 0) s_last[wave] = wl;
        __syncthreads();
        if (tid == 0) a.tw.max_contrib[tile] = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    }
}

hipError_t launch_render_fwd(const RenderFwdArgs& a, hipStream_t s) {
    if (a.grid.tiles == 0) return hipSuccess;
    if (a.store_ctx)
        render_fwd_kernel<true><<<a.grid.tiles, BLOCK, 0, s>>>(a);
    else
        render_fwd_kernel<false><<<a.grid.tiles, BLOCK, 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace exa
