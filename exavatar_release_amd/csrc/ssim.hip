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

}  // namespace exa
