// Per-sub-tile depth sort + front-to-back alpha compositing for gfx950.
//
// ONE WAVE per 8x8-pixel sub-tile (lane l -> pixel (l & 7, l >> 3)); a 256-thread workgroup is four
// independent waves working on four consecutive sub-tiles of one cell.  No __syncthreads anywhere:
// every wave owns its list, a private LDS slice and its 64 pixels, so there is no barrier or
// cross-wave tail to wait for and a finished wave frees its SIMD slot immediately.
//   Phase A: the wave loads its bucket of (depth bits << 32 | id) keys into LDS and sorts it with a
//            wave-synchronous bitonic network (ascending-only comparators: any length, no padding);
//            for training it writes the sorted ids once for the backward pass.
//   Phase B: the sorted list is streamed in batches of 64: each lane gathers ONE 64-byte splat record
//            (48 B used) into the wave's LDS slice, then the 64 pixels blend the batch from LDS
//            broadcast reads, four Gaussians per iteration so four exp2 evaluations are in flight
//            while the serial T recurrence of the previous ones retires.
//
// Replaces upstream SortPairs(depth digit) + renderCUDA (forward) of the rasterizer the reference
// calls at avatar/common/nets/module.py:632-640; per-pixel rule = oracle step 9/10
// (oracle/raster_oracle.py).
//
// Algorithmic HBM bytes: reads 8 B/instance (keys) + 48 B per gathered splat per sub-tile it touches
// (L2-resident after the first touch), writes 4 B/instance (sorted ids, training only) and
// 20 B/pixel (rgb, depth, alpha) + 8 B/pixel (final_T, n_contrib, training only).
#include "common.h"

namespace exa {

constexpr int WAVES = BLOCK / 64;

// Wave-synchronous bitonic sort of k[0..n) ascending.  Only the calling wave touches k.
template <typename KeyPtr>
__device__ __forceinline__ void wave_bitonic_sort_asc(KeyPtr k, int n, int lane) {
    int m = 1;
    while (m < n) m <<= 1;
    const int pairs = m >> 1;
    for (int kk = 2; kk <= m; kk <<= 1) {
        const int half = kk >> 1;
        // flip stage: compare i with its mirror inside each kk-block (both halves ascending)
        for (int i = lane; i < pairs; i += 64) {
            const int off = i & (half - 1);
            const int blk = (i - off) << 1;
            const int lo = blk + off, hi = blk + kk - 1 - off;
            if (hi < n) {
                const unsigned long long a = k[lo], b = k[hi];
                if (a > b) { k[lo] = b; k[hi] = a; }
            }
        }
        wave_lds_fence();
        for (int j = half >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < pairs; i += 64) {
                const int lo = 2 * i - (i & (j - 1)), hi = lo + j;
                if (hi < n) {
                    const unsigned long long a = k[lo], b = k[hi];
                    if (a > b) { k[lo] = b; k[hi] = a; }
                }
            }
            wave_lds_fence();
        }
    }
}

struct PixelState {
    float T, Cr, Cg, Cb, Dp;
    uint32_t last;
    bool done;
};

// Blend one Gaussian (uniform operands g0 = px,py,depth,. ; alpha precomputed per lane) into the pixel.
__device__ __forceinline__ void blend_one(PixelState& s, float alpha, bool valid, const float4& g0, const float4& g2,
                                          uint32_t pos) {
    const float test_T = s.T * (1.0f - alpha);
    const bool stop = valid && !s.done && test_T < T_EPS;
    const bool take = valid && !s.done && !stop;
    const float wgt = take ? alpha * s.T : 0.0f;
    s.Cr = fmaf(g2.x, wgt, s.Cr);
    s.Cg = fmaf(g2.y, wgt, s.Cg);
    s.Cb = fmaf(g2.z, wgt, s.Cb);
    s.Dp = fmaf(g0.z, wgt, s.Dp);
    s.T = take ? test_T : s.T;
    s.last = take ? pos : s.last;
    s.done = s.done || stop;
}

template <bool STORE>
__global__ __launch_bounds__(BLOCK) void render_fwd_kernel(RenderFwdArgs a) {
    __shared__ unsigned long long s_keys[WAVES][SORT_CAP];
    __shared__ float4 s_g0[WAVES][64];
    __shared__ float4 s_g1[WAVES][64];
    __shared__ float4 s_g2[WAVES][64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SubTile sub = decode_subtile(blockIdx.x * WAVES + wave, a.grid, a.tw.cell_order);
    const int st = sub.st;
    if (sub.ox >= a.grid.W || sub.oy >= a.grid.H) return;       // padding sub-tile of a border cell
    const int pxi = sub.ox + (lane & 7), pyi = sub.oy + (lane >> 3);
    const bool inside = pxi < a.grid.W && pyi < a.grid.H;
    const float fx = (float)pxi, fy = (float)pyi;

    const bool overflow = (uint64_t)a.tw.header->num_rendered > a.capacity;
    if (overflow && blockIdx.x == 0 && threadIdx.x == 0) a.tw.header->overflow = 1u;
    const uint2 range = a.tw.ranges[st];
    const int n = overflow ? 0 : (int)(range.y - range.x);

    // ---- phase A: depth sort of this sub-tile's bucket --------------------------------------------
    unsigned long long* gkeys = a.bw.keys + range.x;
    unsigned long long* lkeys = s_keys[wave];
    const bool in_lds = n <= SORT_CAP;
    if (n > 0) {
        if (in_lds) {
            for (int i = lane; i < n; i += 64) lkeys[i] = gkeys[i];
            wave_lds_fence();
            if (n > 1) wave_bitonic_sort_asc(lkeys, n, lane);
            if (STORE)
                for (int i = lane; i < n; i += 64) a.bw.sorted[range.x + i] = (uint32_t)lkeys[i];
        } else {
            // rare: list longer than the LDS slice -> same network on the bucket in global memory
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            wave_bitonic_sort_asc(gkeys, n, lane);
            if (STORE)
                for (int i = lane; i < n; i += 64) a.bw.sorted[range.x + i] = (uint32_t)gkeys[i];
        }
    }

    // ---- phase B: front-to-back blend ----------------------------------------------------------
    PixelState s;
    s.T = 1.0f; s.Cr = 0.f; s.Cg = 0.f; s.Cb = 0.f; s.Dp = 0.f; s.last = 0; s.done = !inside;
    const Splat* __restrict__ splats = a.splats;
    float4* g0s = s_g0[wave];
    float4* g1s = s_g1[wave];
    float4* g2s = s_g2[wave];
    for (int base = 0; base < n; base += 64) {
        if (__all(s.done)) break;
        const int j = base + lane;
        if (j < n) {
            const uint32_t id = in_lds ? (uint32_t)lkeys[j] : (uint32_t)gkeys[j];
            const float4* rec = reinterpret_cast<const float4*>(splats + id);
            g0s[lane] = rec[0];
            g1s[lane] = rec[1];
            g2s[lane] = rec[2];
        }
        wave_lds_fence();
        const int cnt = min(64, n - base);
        int k = 0;
        for (; k + 4 <= cnt; k += 4) {
            float al[4];
            bool va[4];
            float4 q0[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                q0[u] = g0s[k + u];
                const float4 q1 = g1s[k + u];
                const float dx = q0[u].x - fx, dy = q0[u].y - fy;
                const float p2 = gauss_power2(q1.x, q1.y, q1.z, dx, dy);
                al[u] = fminf(ALPHA_MAX, q1.w * gauss_falloff2(p2));
                va[u] = (p2 <= 0.0f) && (al[u] >= ALPHA_MIN);
            }
            if (__any((va[0] || va[1] || va[2] || va[3]) && !s.done)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) blend_one(s, al[u], va[u], q0[u], g2s[k + u], (uint32_t)(base + k + u + 1));
                if (__all(s.done)) break;
            }
        }
        for (; k < cnt; ++k) {
            const float4 q0 = g0s[k];
            const float4 q1 = g1s[k];
            const float dx = q0.x - fx, dy = q0.y - fy;
            const float p2 = gauss_power2(q1.x, q1.y, q1.z, dx, dy);
            const float al = fminf(ALPHA_MAX, q1.w * gauss_falloff2(p2));
            const bool va = (p2 <= 0.0f) && (al >= ALPHA_MIN);
            if (__any(va && !s.done)) blend_one(s, al, va, q0, g2s[k], (uint32_t)(base + k + 1));
        }
        wave_lds_fence();
    }

    // ---- outputs ---------------------------------------------------------------------------------
    const size_t HW = (size_t)a.grid.W * a.grid.H;
    if (inside) {
        const size_t pix = (size_t)pyi * a.grid.W + pxi;
        const float* __restrict__ bg = a.bg;
        a.out_color[pix] = s.Cr + s.T * bg[0];
        a.out_color[HW + pix] = s.Cg + s.T * bg[1];
        a.out_color[2 * HW + pix] = s.Cb + s.T * bg[2];
        a.out_depth[pix] = s.Dp;
        a.out_alpha[pix] = 1.0f - s.T;
        if (STORE) {
            a.iw.final_T[pix] = s.T;
            a.iw.n_contrib[pix] = s.last;
        }
    }
    if (STORE) {
        uint32_t wl = s.last;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) wl = max(wl, (uint32_t)__shfl_xor((int)wl, d, 64));
        if (lane == 0) a.tw.max_contrib[st] = wl;
    }
}

hipError_t launch_render_fwd(const RenderFwdArgs& a, hipStream_t s) {
    if (a.grid.subtiles == 0) return hipSuccess;
    const int blocks = a.grid.subtiles / WAVES;
    if (a.store_ctx)
        render_fwd_kernel<true><<<blocks, BLOCK, 0, s>>>(a);
    else
        render_fwd_kernel<false><<<blocks, BLOCK, 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace exa
