// Fused SSIM map (forward + backward) for gfx950 -- SURVEY.md 8f-4, the image-loss gradient producer that feeds
// render_bwd.
//
// Replaces reference avatar/common/nets/loss.py:31-74 (class SSIM): five grouped 11x11 conv2d calls
// (zero padding 5, Gaussian window sigma 1.5, one group per channel) over img_out, img_target, their squares and
// their product, ~15 elementwise kernels, and the autograd graph of all of it.  Here: one kernel for the map
// (+ the three partial-derivative maps the backward needs), one kernel for dL/d(img_out).
//
//   mu1 = w * x, mu2 = w * y, E11 = w * x^2, E22 = w * y^2, E12 = w * x y          (w = g (x) g, separable)
//   A = 2 mu1 mu2 + C1,  B = 2 (E12 - mu1 mu2) + C2,  C = mu1^2 + mu2^2 + C1,  D = (E11 - mu1^2) + (E22 - mu2^2) + C2
//   ssim = A B / (C D)
// Backward w.r.t. x (the rendered image; the target gets no gradient, as in training):
//   dL/dx = w * (g dssim/dmu1) + 2 x (w * (g dssim/dE11)) + y (w * (g dssim/dE12)),   g = dL/dssim
//   dssim/dmu1 = 2 [ mu2 (B - A) C D - mu1 A B (D - C) ] / (C D)^2     (mu1 also enters sigma1^2 and sigma12)
//   dssim/dE11 = -A B / (C D^2),   dssim/dE12 = 2 A / (C D)
//
// One 16x16 output tile per 256-thread workgroup; the 26x26 input tile (5-pixel halo, zeros outside the image =
// conv2d's zero padding) is staged in LDS, filtered horizontally into LDS, then vertically from LDS.  Pure
// streaming: reads 8 B/pixel, writes 4 (+12 with the derivative maps) B/pixel; backward reads 24, writes 4.
#include "common.h"

namespace exa {

constexpr int ST = 16;                 // output tile
constexpr int SR = 5;                  // window radius (window_size 11)
constexpr int SI = ST + 2 * SR;        // input tile with halo: 26
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

struct SsimWindow { float g[2 * SR + 1]; };

// stage a (SI x SI) halo tile of plane `p` (zeros outside the image)
__device__ __forceinline__ float load_or_zero(const float* __restrict__ p, int x, int y, int W, int H) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? p[(size_t)y * W + x] : 0.0f;
}

__global__ __launch_bounds__(ST * ST) void ssim_fwd_kernel(int H, int W, const float* __restrict__ img1,
                                                           const float* __restrict__ img2, float* __restrict__ map,
                                                           float* __restrict__ dm_dmu1, float* __restrict__ dm_dE11,
                                                           float* __restrict__ dm_dE12, SsimWindow win) {
    __shared__ float s_x[SI][SI + 1], s_y[SI][SI + 1];
    __shared__ float s_h[5][SI][ST + 1];
    const int tid = threadIdx.y * ST + threadIdx.x;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float* __restrict__ p1 = img1 + plane;
    const float* __restrict__ p2 = img2 + plane;
    const int ox = blockIdx.x * ST - SR, oy = blockIdx.y * ST - SR;
    for (int i = tid; i < SI * SI; i += ST * ST) {
        const int ly = i / SI, lx = i - ly * SI;
        s_x[ly][lx] = load_or_zero(p1, ox + lx, oy + ly, W, H);
        s_y[ly][lx] = load_or_zero(p2, ox + lx, oy + ly, W, H);
    }
    __syncthreads();
    for (int i = tid; i < SI * ST; i += ST * ST) {          // horizontal pass: SI rows x ST columns
        const int ly = i / ST, lx = i - ly * ST;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * SR + 1; ++k) {
            const float x = s_x[ly][lx + k], y = s_y[ly][lx + k], wk = win.g[k];
            a = fmaf(wk, x, a); b = fmaf(wk, y, b);
            aa = fmaf(wk, x * x, aa); bb = fmaf(wk, y * y, bb); ab = fmaf(wk, x * y, ab);
        }
        s_h[0][ly][lx] = a; s_h[1][ly][lx] = b; s_h[2][ly][lx] = aa; s_h[3][ly][lx] = bb; s_h[4][ly][lx] = ab;
    }
    __syncthreads();
    const int px = blockIdx.x * ST + threadIdx.x, py = blockIdx.y * ST + threadIdx.y;
    if (px >= W || py >= H) return;
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 2 * SR + 1; ++k) {                  // vertical pass
        const float wk = win.g[k];
        mu1 = fmaf(wk, s_h[0][threadIdx.y + k][threadIdx.x], mu1);
        mu2 = fmaf(wk, s_h[1][threadIdx.y + k][threadIdx.x], mu2);
        e11 = fmaf(wk, s_h[2][threadIdx.y + k][threadIdx.x], e11);
        e22 = fmaf(wk, s_h[3][threadIdx.y + k][threadIdx.x], e22);
        e12 = fmaf(wk, s_h[4][threadIdx.y + k][threadIdx.x], e12);
    }
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
    const float A = 2.0f * mu12 + SSIM_C1, B = 2.0f * s12 + SSIM_C2;
    const float C = mu1_sq + mu2_sq + SSIM_C1, D = s1 + s2 + SSIM_C2;
    const float inv_CD = 1.0f / (C * D);
    const size_t o = plane + (size_t)py * W + px;
    map[o] = A * B * inv_CD;
    if (dm_dmu1) {
        dm_dmu1[o] = 2.0f * (mu2 * (B - A) * C * D - mu1 * A * B * (D - C)) * inv_CD * inv_CD;
        dm_dE11[o] = -A * B * inv_CD / D;
        dm_dE12[o] = 2.0f * A * inv_CD;
    }
}

__global__ __launch_bounds__(ST * ST) void ssim_bwd_kernel(int H, int W, const float* __restrict__ img1,
                                                           const float* __restrict__ img2, const float* __restrict__ dL_dmap,
                                                           const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dE11,
                                                           const float* __restrict__ dm_dE12, float* __restrict__ dL_dimg1,
                                                           SsimWindow win) {
    __shared__ float s_in[3][SI][SI + 1];
    __shared__ float s_h[3][SI][ST + 1];
    const int tid = threadIdx.y * ST + threadIdx.x;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int ox = blockIdx.x * ST - SR, oy = blockIdx.y * ST - SR;
    for (int i = tid; i < SI * SI; i += ST * ST) {
        const int ly = i / SI, lx = i - ly * SI;
        const int x = ox + lx, y = oy + ly;
        float g = 0.f, a = 0.f, b = 0.f, c = 0.f;
        if (x >= 0 && x < W && y >= 0 && y < H) {
            const size_t o = plane + (size_t)y * W + x;
            g = dL_dmap[o]; a = dm_dmu1[o]; b = dm_dE11[o]; c = dm_dE12[o];
        }
        s_in[0][ly][lx] = g * a; s_in[1][ly][lx] = g * b; s_in[2][ly][lx] = g * c;
    }
    __syncthreads();
    for (int i = tid; i < SI * ST; i += ST * ST) {
        const int ly = i / ST, lx = i - ly * ST;
        float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * SR + 1; ++k) {
            const float wk = win.g[k];
            a = fmaf(wk, s_in[0][ly][lx + k], a); b = fmaf(wk, s_in[1][ly][lx + k], b); c = fmaf(wk, s_in[2][ly][lx + k], c);
        }
        s_h[0][ly][lx] = a; s_h[1][ly][lx] = b; s_h[2][ly][lx] = c;
    }
    __syncthreads();
    const int px = blockIdx.x * ST + threadIdx.x, py = blockIdx.y * ST + threadIdx.y;
    if (px >= W || py >= H) return;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int k = 0; k < 2 * SR + 1; ++k) {
        const float wk = win.g[k];
        c0 = fmaf(wk, s_h[0][threadIdx.y + k][threadIdx.x], c0);
        c1 = fmaf(wk, s_h[1][threadIdx.y + k][threadIdx.x], c1);
        c2 = fmaf(wk, s_h[2][threadIdx.y + k][threadIdx.x], c2);
    }
    const size_t o = plane + (size_t)py * W + px;
    dL_dimg1[o] = c0 + 2.0f * img1[o] * c1 + img2[o] * c2;
}

// The window exactly as the reference builds it (loss.py:35-37): exp() in double, rounded to float32, normalised
// by its float32 sum.
static SsimWindow make_window() {
    SsimWindow w;
    float sum = 0.f;
    for (int i = 0; i < 2 * SR + 1; ++i) {
        w.g[i] = (float)exp(-(double)((i - SR) * (i - SR)) / (2.0 * 1.5 * 1.5));
        sum += w.g[i];
    }
    for (int i = 0; i < 2 * SR + 1; ++i) w.g[i] /= sum;
    return w;
}

// =====================================================================================================================
// Fused photometric loss: the weighted L1 + (1 - SSIM) objective of one render -- reference
// avatar/main/model.py:197-198,204-205,214-215 with the classes of avatar/common/nets/loss.py:11-74 -- evaluated in TWO
// kernels that hand render_bwd its dL/d(image) directly:
//     loss = w_l1 * mean(l1w * |x - y|) + w_ssim * mean(1 - ssim(x * m, y * m))        over the bbox crop
// photo_stats_kernel: SSIM statistics of a 32x32 output tile (42x42 halo tile in LDS, separable 11-tap passes with
//     four outputs per thread so that every LDS value feeds four sums), the three partial-derivative maps, and this
//     tile's partial sums of the SSIM map and of the L1 term (plain stores: deterministic, no atomics).
// photo_grad_kernel: the second 11x11 pass over the derivative maps + the L1 sign term, scaled by the loss weights and
//     the 1 / N of the two means -> dL/dx, written once.  The crop window is the image as far as SSIM's zero padding is
//     concerned (the reference crops BEFORE the convolutions, loss.py:50-58).
// Replaces, per render: 5 grouped conv2d + ~25 elementwise / reduction kernels forward and their autograd graph.
// HBM bytes per pixel and channel: stats reads 8 (+ masks) and writes 12, grad reads 12 + 8 and writes 4.
constexpr int PT = 32;                 // output tile edge
constexpr int PI = PT + 2 * SR;        // 42: input tile edge with halo
constexpr int PBLOCK2 = 256;

struct PhotoArgs {
    int C, H, W;                       // channels per image, full image size
    int cx0, cy0, cw, ch;              // crop window (clamped bbox) = the image the SSIM convolutions see
    const float* x; const float* y;    // [B, C, H, W]
    const float* l1w;                  // [B, 1, H, W] weight of the L1 term, or NULL (= 1)
    const float* smask;                // [B, 1, H, W] mask multiplied into both images before SSIM, or NULL
    float* maps;                       // [3][B * C][ch][cw] derivative maps
    float* partials;                   // [blocks][2]: sum of ssim, sum of l1w |x - y|
    float k_l1, k_ssim;                // w_l1 / N and w_ssim / N
    float* dL_dx;                      // [B, C, H, W] (only the crop window is written)
};

__global__ __launch_bounds__(PBLOCK2) void photo_stats_kernel(PhotoArgs a, SsimWindow win) {
    // LDS: the two halo tiles (14.4 KB) + the horizontally filtered statistics of ONE GROUP at a time (round 6: {mu1, mu2}, then
    // {E11, E22, E12}: 16.6 KB instead of 27.7 for all five) = 31 KB per workgroup: five workgroups per CU instead of three.  The
    // kernel is a chain of barrier-separated phases (load, horizontal, vertical): what it lacked was waves to overlap them.
    __shared__ float s_x[PI][PI + 1], s_y[PI][PI + 1];
    __shared__ float s_h[3][PI][PT + 1];
    __shared__ float s_red[2][PBLOCK2 / 64];
    const int tid = threadIdx.x;
    const int n = blockIdx.z;                                   // plane = image * C + channel
    const size_t plane = (size_t)n * a.H * a.W, mplane = (size_t)(n / a.C) * a.H * a.W;
    const float* __restrict__ px = a.x + plane;
    const float* __restrict__ py = a.y + plane;
    const float* __restrict__ pm = a.smask ? a.smask + mplane : nullptr;
    const int ox = blockIdx.x * PT - SR, oy = blockIdx.y * PT - SR;
    // The halo tile: ALL of a thread's loads leave before the first one is waited for (addresses clamped into the crop window, the
    // value zeroed afterwards where the tap lies outside: conv2d's zero padding) -- a load and the LDS store of its result in one
    // loop body had compiled to seven serial round trips per workgroup.
    {
        constexpr int NLD = (PI * PI + PBLOCK2 - 1) / PBLOCK2;
        float vx[NLD], vy[NLD], vm[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = min(tid + u * PBLOCK2, PI * PI - 1);
            const int ly = i / PI, lx = i - ly * PI;
            const int gx = min(max(ox + lx, 0), a.cw - 1), gy = min(max(oy + ly, 0), a.ch - 1);
            const size_t o = (size_t)(a.cy0 + gy) * a.W + (a.cx0 + gx);
            vx[u] = px[o]; vy[u] = py[o];
            vm[u] = pm ? pm[o] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + u * PBLOCK2;
            if (i < PI * PI) {
                const int ly = i / PI, lx = i - ly * PI;
                const bool in = ox + lx >= 0 && ox + lx < a.cw && oy + ly >= 0 && oy + ly < a.ch;
                s_x[ly][lx] = in ? (pm ? vx[u] * vm[u] : vx[u]) : 0.0f;
                s_y[ly][lx] = in ? (pm ? vy[u] * vm[u] : vy[u]) : 0.0f;
            }
        }
    }
    __syncthreads();
    const int tx = tid & (PT - 1), ty = tid / PT;               // vertical pass: thread = (column tx, rows 4 ty .. 4 ty + 3)
    float acc[4][5];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[j][q] = 0.f;
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        // horizontal pass: PI rows x PT columns in ONE sweep of the workgroup -- thread = (row tid % 42, column chunk tid / 42): six
        // chunks of 6, 6, 5, 5, 5, 5 columns, 16 LDS values feed up to 6 x 11 taps (four columns per item had left the second of two
        // sweeps with 80 of 256 threads at work); lanes of a wave read different rows: stride 43 words, conflict-free
        if (tid < 6 * PI) {
            const int ly = tid % PI, ch = tid / PI;
            const int c0 = ch < 2 ? 6 * ch : 12 + 5 * (ch - 2), nc = ch < 2 ? 6 : 5;
            float xv[16], yv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { xv[k] = s_x[ly][min(c0 + k, PI - 1)]; yv[k] = s_y[ly][min(c0 + k, PI - 1)]; }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (j >= nc) break;
                if (grp == 0) {
                    float m1 = 0.f, m2 = 0.f;
#pragma unroll
                    for (int k = 0; k < 2 * SR + 1; ++k) {
                        const float wk = win.g[k];
                        m1 = fmaf(wk, xv[j + k], m1); m2 = fmaf(wk, yv[j + k], m2);
                    }
                    s_h[0][ly][c0 + j] = m1; s_h[1][ly][c0 + j] = m2;
                } else {
                    float e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
                    for (int k = 0; k < 2 * SR + 1; ++k) {
                        const float wk = win.g[k], xx = xv[j + k], yy = yv[j + k];
                        e11 = fmaf(wk, xx * xx, e11); e22 = fmaf(wk, yy * yy, e22); e12 = fmaf(wk, xx * yy, e12);
                    }
                    s_h[0][ly][c0 + j] = e11; s_h[1][ly][c0 + j] = e22; s_h[2][ly][c0 + j] = e12;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < (grp == 0 ? 2 : 3); ++q) {
            float v[14];
#pragma unroll
            for (int k = 0; k < 14; ++k) v[k] = s_h[q][ty * 4 + k][tx];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 2 * SR + 1; ++k) acc[j][grp * 2 + q] = fmaf(win.g[k], v[j + k], acc[j][grp * 2 + q]);
        }
        if (grp == 0) __syncthreads();                          // the second group overwrites s_h
    }
    float sum_ssim = 0.f, sum_l1 = 0.f;
    const int lx = blockIdx.x * PT + tx;
    const size_t map_plane = (size_t)a.cw * a.ch, n_planes = (size_t)gridDim.z;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ly = blockIdx.y * PT + ty * 4 + j;
        if (lx >= a.cw || ly >= a.ch) continue;
        const float mu1 = acc[j][0], mu2 = acc[j][1], e11 = acc[j][2], e22 = acc[j][3], e12 = acc[j][4];
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
        const float A = 2.0f * mu12 + SSIM_C1, B = 2.0f * s12 + SSIM_C2;
        const float Cc = mu1_sq + mu2_sq + SSIM_C1, D = s1 + s2 + SSIM_C2;
        const float inv_CD = 1.0f / (Cc * D);
        sum_ssim += A * B * inv_CD;
        const size_t o = (size_t)n * map_plane + (size_t)ly * a.cw + lx;
        a.maps[o] = 2.0f * (mu2 * (B - A) * Cc * D - mu1 * A * B * (D - Cc)) * inv_CD * inv_CD;
        a.maps[n_planes * map_plane + o] = -A * B * inv_CD / D;
        a.maps[2 * n_planes * map_plane + o] = 2.0f * A * inv_CD;
        const size_t g = (size_t)(a.cy0 + ly) * a.W + (a.cx0 + lx);
        const float d = fabsf(px[g] - py[g]);
        sum_l1 += a.l1w ? d * a.l1w[mplane + g] : d;
    }
    // deterministic block reduction -> one pair of partial sums per workgroup
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { sum_ssim += __shfl_xor(sum_ssim, d, 64); sum_l1 += __shfl_xor(sum_l1, d, 64); }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = sum_ssim; s_red[1][tid >> 6] = sum_l1; }
    __syncthreads();
    if (tid == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        a.partials[2 * b] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
        a.partials[2 * b + 1] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
    }
}

__global__ __launch_bounds__(PBLOCK2) void photo_grad_kernel(PhotoArgs a, SsimWindow win) {
    // (round 6) the horizontally filtered maps go through LDS ONE at a time: 21.7 + 5.5 KB instead of 21.7 + 16.6 per workgroup, five
    // workgroups per CU instead of four
    __shared__ float s_in[3][PI][PI + 1];
    __shared__ float s_h[PI][PT + 1];
    const int tid = threadIdx.x;
    const int n = blockIdx.z;
    const size_t plane = (size_t)n * a.H * a.W, mplane = (size_t)(n / a.C) * a.H * a.W;
    const size_t map_plane = (size_t)a.cw * a.ch, n_planes = (size_t)gridDim.z;
    const int ox = blockIdx.x * PT - SR, oy = blockIdx.y * PT - SR;
    {   // all loads of the halo tiles up front, unconditional (clamped addresses, zeroed outside the window): see photo_stats_kernel
        constexpr int NLD = (PI * PI + PBLOCK2 - 1) / PBLOCK2;
        float v0[NLD], v1[NLD], v2[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = min(tid + u * PBLOCK2, PI * PI - 1);
            const int ly = i / PI, lx = i - ly * PI;
            const int x = min(max(ox + lx, 0), a.cw - 1), y = min(max(oy + ly, 0), a.ch - 1);
            const size_t o = (size_t)n * map_plane + (size_t)y * a.cw + x;
            v0[u] = a.maps[o]; v1[u] = a.maps[n_planes * map_plane + o]; v2[u] = a.maps[2 * n_planes * map_plane + o];
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + u * PBLOCK2;
            if (i < PI * PI) {
                const int ly = i / PI, lx = i - ly * PI;
                const bool in = ox + lx >= 0 && ox + lx < a.cw && oy + ly >= 0 && oy + ly < a.ch;
                s_in[0][ly][lx] = in ? v0[u] : 0.f; s_in[1][ly][lx] = in ? v1[u] : 0.f; s_in[2][ly][lx] = in ? v2[u] : 0.f;
            }
        }
    }
    __syncthreads();
    const int tx = tid & (PT - 1), ty = tid / PT;
    float acc[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j][0] = 0.f; acc[j][1] = 0.f; acc[j][2] = 0.f; }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (tid < 6 * PI) {                                     // one sweep: thread = (row, column chunk), as in photo_stats_kernel
            const int ly = tid % PI, ch = tid / PI;
            const int c0 = ch < 2 ? 6 * ch : 12 + 5 * (ch - 2), nc = ch < 2 ? 6 : 5;
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = s_in[q][ly][min(c0 + k, PI - 1)];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (j >= nc) break;
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < 2 * SR + 1; ++k) sum = fmaf(win.g[k], v[j + k], sum);
                s_h[ly][c0 + j] = sum;
            }
        }
        __syncthreads();
        float v[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) v[k] = s_h[ty * 4 + k][tx];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 2 * SR + 1; ++k) acc[j][q] = fmaf(win.g[k], v[j + k], acc[j][q]);
        if (q < 2) __syncthreads();                             // the next map overwrites s_h
    }
    const int lx = blockIdx.x * PT + tx;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ly = blockIdx.y * PT + ty * 4 + j;
        if (lx >= a.cw || ly >= a.ch) continue;
        const size_t g = (size_t)(a.cy0 + ly) * a.W + (a.cx0 + lx);
        const float xr = a.x[plane + g], yr = a.y[plane + g];
        const float m = a.smask ? a.smask[mplane + g] : 1.0f;
        // d mean(1 - ssim(x m, y m)) / dx = -(1/N) m [c0 + 2 (x m) c1 + (y m) c2]
        const float gs = -a.k_ssim * m * (acc[j][0] + 2.0f * (xr * m) * acc[j][1] + (yr * m) * acc[j][2]);
        const float d = xr - yr;
        const float sgn = (d > 0.0f ? 1.0f : 0.0f) - (d < 0.0f ? 1.0f : 0.0f);
        const float gl = a.k_l1 * sgn * (a.l1w ? a.l1w[mplane + g] : 1.0f);
        a.dL_dx[plane + g] = gs + gl;
    }
}

// L1 map of reference RGBLoss (avatar/common/nets/loss.py:11-29) as one kernel: |x - t| over the crop window with
// t = y * mask + (1 - mask) * bg when a mask and a background colour are given; and its backward sign(x - t) * g.
struct L1Args {
    int C, H, W, cx0, cy0, cw, ch;
    const float* x; const float* y; const float* mask; const float* bg;       // mask [B,1,H,W], bg [B,C] or NULL
    const float* g; float* out;                                                 // forward: out = map [B,C,ch,cw];
};                                                                              // backward: g = dL/dmap, out = dL/dx [B,C,H,W]
template <bool BWD>
__global__ __launch_bounds__(BLOCK) void l1_kernel(L1Args a, size_t total) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= total) return;
    const int lx = (int)(i % a.cw), ly = (int)((i / a.cw) % a.ch);
    const size_t n = i / ((size_t)a.cw * a.ch);                 // plane = image * C + channel
    const size_t g = (size_t)(a.cy0 + ly) * a.W + (a.cx0 + lx);
    const size_t o = n * a.H * a.W + g;
    float t = a.y[o];
    if (a.mask && a.bg) {
        const float m = a.mask[(n / a.C) * (size_t)a.H * a.W + g];
        t = t * m + (1.0f - m) * a.bg[n];
    }
    const float d = a.x[o] - t;
    if (BWD) a.out[o] = ((d > 0.0f ? 1.0f : 0.0f) - (d < 0.0f ? 1.0f : 0.0f)) * a.g[i];
    else a.out[i] = fabsf(d);
}

hipError_t launch_ssim_fwd(int N, int H, int W, const float* img1, const float* img2, float* map, float* dm_dmu1,
                           float* dm_dE11, float* dm_dE12, hipStream_t s) {
    if (N == 0 || H == 0 || W == 0) return hipSuccess;
    const dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, N), block(ST, ST, 1);
    ssim_fwd_kernel<<<grid, block, 0, s>>>(H, W, img1, img2, map, dm_dmu1, dm_dE11, dm_dE12, make_window());
    return hipGetLastError();
}

hipError_t launch_ssim_bwd(int N, int H, int W, const float* img1, const float* img2, const float* dL_dmap,
                           const float* dm_dmu1, const float* dm_dE11, const float* dm_dE12, float* dL_dimg1,
                           hipStream_t s) {
    if (N == 0 || H == 0 || W == 0) return hipSuccess;
    const dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, N), block(ST, ST, 1);
    ssim_bwd_kernel<<<grid, block, 0, s>>>(H, W, img1, img2, dL_dmap, dm_dmu1, dm_dE11, dm_dE12, dL_dimg1, make_window());
    return hipGetLastError();
}

hipError_t launch_photo_loss(int B, int C, int H, int W, const int* crop, const float* x, const float* y, const float* l1w,
                             const float* smask, float w_l1, float w_ssim, float* maps, float* partials, float* dL_dx,
                             int stage, hipStream_t s) {
    PhotoArgs a;
    a.C = C; a.H = H; a.W = W; a.cx0 = crop[0]; a.cy0 = crop[1]; a.cw = crop[2]; a.ch = crop[3];
    if (B * C == 0 || a.cw <= 0 || a.ch <= 0) return hipSuccess;
    a.x = x; a.y = y; a.l1w = l1w; a.smask = smask; a.maps = maps; a.partials = partials; a.dL_dx = dL_dx;
    const float inv_n = 1.0f / ((float)B * (float)C * (float)a.cw * (float)a.ch);
    a.k_l1 = w_l1 * inv_n; a.k_ssim = w_ssim * inv_n;
    const dim3 grid((a.cw + PT - 1) / PT, (a.ch + PT - 1) / PT, B * C);
    if (stage == 0) photo_stats_kernel<<<grid, PBLOCK2, 0, s>>>(a, make_window());
    else photo_grad_kernel<<<grid, PBLOCK2, 0, s>>>(a, make_window());
    return hipGetLastError();
}

hipError_t launch_l1(int B, int C, int H, int W, const int* crop, const float* x, const float* y, const float* mask,
                     const float* bg, const float* g, float* out, int backward, hipStream_t s) {
    L1Args a;
    a.C = C; a.H = H; a.W = W; a.cx0 = crop[0]; a.cy0 = crop[1]; a.cw = crop[2]; a.ch = crop[3];
    a.x = x; a.y = y; a.mask = mask; a.bg = bg; a.g = g; a.out = out;
    if (a.cw <= 0 || a.ch <= 0) return hipSuccess;
    const size_t total = (size_t)B * C * a.cw * a.ch;
    if (total == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((total + BLOCK - 1) / BLOCK);
    if (backward) l1_kernel<true><<<blocks, BLOCK, 0, s>>>(a, total);
    else l1_kernel<false><<<blocks, BLOCK, 0, s>>>(a, total);
    return hipGetLastError();
}

}  // namespace exa
