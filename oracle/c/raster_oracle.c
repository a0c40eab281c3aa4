/*
 * raster_oracle.c -- second, independent CPU restatement (plain C99 + OpenMP) of the differentiable 3D-Gaussian
 * rasterizer behind ExAvatar's GaussianRenderer.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may load it (through oracle/c_oracle.py); the product path never does.
 *
 * PARITY UNPINNED, like oracle/raster_oracle.py: the rasterizer's source (pip module
 * `diff_gaussian_rasterization_depth`, reference avatar/common/nets/module.py:11, avatar/README.md:42) is not in
 * the reference tree and the reference holds no test or golden vector for it.  This file
 * restates the published algorithm of graphdeco-inria/diff-gaussian-rasterization (+ the depth / alpha outputs of
 * the -depth fork) in the SHAPE upstream has it -- one sequential loop per pixel front to back, one loop per pixel
 * back to front for the gradients, a hand-written chain rule per Gaussian (upstream's computeCov2D / preprocess
 * backward, incl. `x_grad_mul` and `1 / (det^2 + 1e-7)`) -- whereas oracle/raster_oracle.py is vectorised PyTorch
 * with autograd.  The two share no code; tests/test_c_oracle.py holds them against each other and against the
 * committed golden vectors.
 *
 * Conventions: call site reference module.py:609-640; camera matrices as module.py:604-608 builds them (row-vector
 * convention: p_view = [mu, 1] @ viewmatrix).  Forward arithmetic is float32 in the association order of
 * oracle/raster_oracle.py::preprocess (compile with -ffp-contract=off), so radii, tile rects and per-tile orders
 * are bit-identical to it; the backward accumulates in double.
 *
 * Steps: 1 cull p_view.z <= 0.2 | 2 NDC with +1e-7 | 3 Sigma3 = R S^2 R^T, q = (w,x,y,z) as is |
 * 4 EWA with the +-1.3 tanfov clamp, +0.3 low-pass, conic | 5 radius = ceil(3 sqrt(mid + sqrt(max(0.1, mid^2 - det)))) |
 * 6 pix = ((ndc + 1) size - 1) / 2 | 7 16x16 tile rect, truncation, clamp | 8 per tile ascending (depth bits, index) |
 * 9 per pixel: skip power > 0, alpha = min(0.99, o exp(power)), skip alpha < 1/255, stop (without blending) when
 * T (1 - alpha) < 1e-4 | 10 color = C + T bg, depth = sum z alpha T, alpha = 1 - T.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16
#define NEAR_CULL 0.2f
#define LOWPASS 0.3f
#define ALPHA_MAX 0.99f
#define ALPHA_MIN (1.0f / 255.0f)
#define T_EPS 1e-4f

typedef struct {
    int32_t image_height, image_width;
    float tanfovx, tanfovy, scale_modifier;
    int32_t sh_degree;
    float bg[3];
    float viewmatrix[16];   /* row-major of the [4,4] tensor the reference passes (module.py:605) */
    float projmatrix[16];   /* row-major of view^T @ proj^T (module.py:607) */
    float campos[3];
} OrcSettings;

/* SH constants: reference avatar/common/utils/transforms.py:82-110 */
static const double SH_C0 = 0.28209479177387814, SH_C1 = 0.4886025119029199;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
                                0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                                -0.4570457994644658, 1.445305721320277, -0.5900435899266435};

typedef struct {
    float px, py, depth, A, B, C, opacity, col[3];
    int radius, x0, y0, x1, y1;
    unsigned clamped;        /* bit c: SH colour channel c clamped at 0 */
} Geo;

typedef struct { uint32_t depth_bits; int32_t id; } Key;

static int key_cmp(const void* a, const void* b) {
    const Key* x = (const Key*)a; const Key* y = (const Key*)b;
    if (x->depth_bits != y->depth_bits) return x->depth_bits < y->depth_bits ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

static int clampi(long v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int)v); }
static long trunc_clamped(float t) {           /* C float -> int truncation, NaN -> 0, clamped to +-2^20 */
    if (t != t) return 0;
    if (t > 1048576.0f) t = 1048576.0f;
    if (t < -1048576.0f) t = -1048576.0f;
    return (long)t;
}

/* SH basis of degree <= 3 at unit direction (x, y, z): transforms.py:112-167 */
static void sh_basis(int deg, double x, double y, double z, double* b) {
    for (int i = 0; i < 16; ++i) b[i] = 0.0;
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
        if (deg > 1) {
            const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (2.0 * zz - xx - yy);
            b[7] = SH_C2[3] * xz; b[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = SH_C3[0] * y * (3.0 * xx - yy); b[10] = SH_C3[1] * xy * z;
                b[11] = SH_C3[2] * y * (4.0 * zz - xx - yy); b[12] = SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy);
                b[13] = SH_C3[4] * x * (4.0 * zz - xx - yy); b[14] = SH_C3[5] * z * (xx - yy);
                b[15] = SH_C3[6] * x * (xx - 3.0 * yy);
            }
        }
    }
}

/* float32 SH colour in the association order of oracle/raster_oracle.py::eval_sh_color (module.py:258-266) */
static void sh_colour_f32(const OrcSettings* s, const float* sh /* [M][3] */, const float* mu, float* col,
                          unsigned* clamped) {
    const float C0 = (float)SH_C0, C1 = (float)SH_C1;
    float dx = mu[0] - s->campos[0], dy = mu[1] - s->campos[1], dz = mu[2] - s->campos[2];
    const float n = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float x = dx / n, y = dy / n, z = dz / n;
    *clamped = 0u;
    for (int c = 0; c < 3; ++c) {
        float res = C0 * sh[0 * 3 + c];
        if (s->sh_degree > 0) {
            res = ((res - (C1 * y) * sh[1 * 3 + c]) + (C1 * z) * sh[2 * 3 + c]) - (C1 * x) * sh[3 * 3 + c];
            if (s->sh_degree > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = ((((res + ((float)SH_C2[0] * xy) * sh[4 * 3 + c]) + ((float)SH_C2[1] * yz) * sh[5 * 3 + c]) +
                        ((float)SH_C2[2] * ((2.0f * zz - xx) - yy)) * sh[6 * 3 + c]) +
                       ((float)SH_C2[3] * xz) * sh[7 * 3 + c]) + ((float)SH_C2[4] * (xx - yy)) * sh[8 * 3 + c];
                if (s->sh_degree > 2) {
                    res = ((((((res + (((float)SH_C3[0] * y) * (3.0f * xx - yy)) * sh[9 * 3 + c]) +
                               (((float)SH_C3[1] * xy) * z) * sh[10 * 3 + c]) +
                              (((float)SH_C3[2] * y) * ((4.0f * zz - xx) - yy)) * sh[11 * 3 + c]) +
                             (((float)SH_C3[3] * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy)) * sh[12 * 3 + c]) +
                            (((float)SH_C3[4] * x) * ((4.0f * zz - xx) - yy)) * sh[13 * 3 + c]) +
                           (((float)SH_C3[5] * z) * (xx - yy)) * sh[14 * 3 + c]) +
                          (((float)SH_C3[6] * x) * (xx - 3.0f * yy)) * sh[15 * 3 + c];
                }
            }
        }
        res = res + 0.5f;
        if (res < 0.0f) { res = 0.0f; *clamped |= 1u << c; }
        col[c] = res;
    }
}

/* steps 1-7 for one Gaussian, float32, every * and + separately rounded in the order of raster_oracle.py::preprocess */
static void preprocess_one(const OrcSettings* s, int idx, const float* means3D, const float* opacities,
                           const float* scales, const float* rotations, const float* cov3D, int gx, int gy, Geo* g) {
    const float* v = s->viewmatrix; const float* p = s->projmatrix;
    const int W = s->image_width, H = s->image_height;
    const float x = means3D[idx * 3], y = means3D[idx * 3 + 1], z = means3D[idx * 3 + 2];
    memset(g, 0, sizeof(*g));
    const float pvx = ((v[0] * x + v[4] * y) + v[8] * z) + v[12];
    const float pvy = ((v[1] * x + v[5] * y) + v[9] * z) + v[13];
    const float pvz = ((v[2] * x + v[6] * y) + v[10] * z) + v[14];
    const float hx = ((p[0] * x + p[4] * y) + p[8] * z) + p[12];
    const float hy = ((p[1] * x + p[5] * y) + p[9] * z) + p[13];
    const float hw = ((p[3] * x + p[7] * y) + p[11] * z) + p[15];
    const float pw = 1.0f / (hw + 1e-7f);
    const float ndcx = hx * pw, ndcy = hy * pw;
    float S00, S01, S02, S11, S12, S22;
    if (cov3D) {
        const float* c6 = cov3D + idx * 6;
        S00 = c6[0]; S01 = c6[1]; S02 = c6[2]; S11 = c6[3]; S12 = c6[4]; S22 = c6[5];
    } else {
        const float s0 = s->scale_modifier * scales[idx * 3], s1 = s->scale_modifier * scales[idx * 3 + 1],
                    s2 = s->scale_modifier * scales[idx * 3 + 2];
        const float qr = rotations[idx * 4], qx = rotations[idx * 4 + 1], qy = rotations[idx * 4 + 2],
                    qz = rotations[idx * 4 + 3];
        const float R00 = 1.0f - 2.0f * (qy * qy + qz * qz), R01 = 2.0f * (qx * qy - qr * qz), R02 = 2.0f * (qx * qz + qr * qy);
        const float R10 = 2.0f * (qx * qy + qr * qz), R11 = 1.0f - 2.0f * (qx * qx + qz * qz), R12 = 2.0f * (qy * qz - qr * qx);
        const float R20 = 2.0f * (qx * qz - qr * qy), R21 = 2.0f * (qy * qz + qr * qx), R22 = 1.0f - 2.0f * (qx * qx + qy * qy);
        const float M00 = R00 * s0, M01 = R01 * s1, M02 = R02 * s2;
        const float M10 = R10 * s0, M11 = R11 * s1, M12 = R12 * s2;
        const float M20 = R20 * s0, M21 = R21 * s1, M22 = R22 * s2;
        S00 = (M00 * M00 + M01 * M01) + M02 * M02;
        S01 = (M00 * M10 + M01 * M11) + M02 * M12;
        S02 = (M00 * M20 + M01 * M21) + M02 * M22;
        S11 = (M10 * M10 + M11 * M11) + M12 * M12;
        S12 = (M10 * M20 + M11 * M21) + M12 * M22;
        S22 = (M20 * M20 + M21 * M21) + M22 * M22;
    }
    const float focal_x = (float)W / (2.0f * s->tanfovx), focal_y = (float)H / (2.0f * s->tanfovy);
    const float limx = 1.3f * s->tanfovx, limy = 1.3f * s->tanfovy;
    const float tz = pvz;
    const float tx = fminf(limx, fmaxf(-limx, pvx / tz)) * tz;
    const float ty = fminf(limy, fmaxf(-limy, pvy / tz)) * tz;
    const float J00 = focal_x / tz, J02 = -(focal_x * tx) / (tz * tz);
    const float J11 = focal_y / tz, J12 = -(focal_y * ty) / (tz * tz);
    const float T00 = J00 * v[0] + J02 * v[2], T01 = J00 * v[4] + J02 * v[6], T02 = J00 * v[8] + J02 * v[10];
    const float T10 = J11 * v[1] + J12 * v[2], T11 = J11 * v[5] + J12 * v[6], T12 = J11 * v[9] + J12 * v[10];
    const float U00 = (T00 * S00 + T01 * S01) + T02 * S02, U01 = (T00 * S01 + T01 * S11) + T02 * S12,
                U02 = (T00 * S02 + T01 * S12) + T02 * S22;
    const float U10 = (T10 * S00 + T11 * S01) + T12 * S02, U11 = (T10 * S01 + T11 * S11) + T12 * S12,
                U12 = (T10 * S02 + T11 * S12) + T12 * S22;
    const float a = ((U00 * T00 + U01 * T01) + U02 * T02) + LOWPASS;
    const float b = (U00 * T10 + U01 * T11) + U02 * T12;
    const float c = ((U10 * T10 + U11 * T11) + U12 * T12) + LOWPASS;
    const float det = a * c - b * b;
    const float det_inv = 1.0f / (det == 0.0f ? 1.0f : det);
    const float mid = 0.5f * (a + c);
    const float lam = mid + sqrtf(fmaxf(mid * mid - det, 0.1f));
    const float r_f = 3.0f * sqrtf(lam);
    /* ceil of a NaN / inf radius: treated as 0 rows below (the rect turns empty or the cull fires) */
    const float rc = ceilf(r_f);
    long radius = (rc == rc && rc < 9.0e18f) ? (long)rc : 0;
    if (radius > 2147483647L) radius = 2147483647L;
    const float px = ((ndcx + 1.0f) * (float)W - 1.0f) * 0.5f;
    const float py = ((ndcy + 1.0f) * (float)H - 1.0f) * 0.5f;
    const float rf = (float)radius;
    const int x0 = clampi(trunc_clamped((px - rf) / TILE), 0, gx), x1 = clampi(trunc_clamped(((px + rf) + (TILE - 1)) / TILE), 0, gx);
    const int y0 = clampi(trunc_clamped((py - rf) / TILE), 0, gy), y1 = clampi(trunc_clamped(((py + rf) + (TILE - 1)) / TILE), 0, gy);
    const int tiles = (x1 - x0) * (y1 - y0);
    const int vis = (tz > NEAR_CULL) && (det != 0.0f) && tiles > 0;
    g->px = px; g->py = py; g->depth = tz;
    g->A = c * det_inv; g->B = -b * det_inv; g->C = a * det_inv;
    g->opacity = opacities[idx];
    if (vis) { g->radius = (int)radius; g->x0 = x0; g->y0 = y0; g->x1 = x1; g->y1 = y1; }
}

/* per-pixel falloff, float32, in the order of raster_oracle.py::rasterize */
static inline float gauss_power(const Geo* g, float fx, float fy, float* dx, float* dy) {
    *dx = g->px - fx; *dy = g->py - fy;
    return -0.5f * ((g->A * *dx) * *dx + (g->C * *dy) * *dy) - (g->B * *dx) * *dy;
}

/* per-Gaussian chain rule in double: accumulators acc = (dL/dpx, dL/dpy, dL/dA, dL/dB, dL/dC, dL/dopacity,
 * dL/dcolour[3], dL/ddepth) -> gradients of the inputs.  Upstream computeCov2DCUDA + preprocessCUDA (backward). */
static void preprocess_backward_one(const OrcSettings* s, int idx, int sh_M, const float* means3D, const float* shs,
                                    const float* scales, const float* rotations, const float* cov3D, const Geo* g,
                                    const double* acc, float* dmeans2D, float* dmeans3D, float* dcolors, float* dopacity,
                                    float* dscales, float* drot, float* dsh, float* dcov3D) {
    const float* vf = s->viewmatrix; const float* pf = s->projmatrix;
    double v[16], p[16];
    for (int i = 0; i < 16; ++i) { v[i] = vf[i]; p[i] = pf[i]; }
    const int W = s->image_width, H = s->image_height;
    const double x = means3D[idx * 3], y = means3D[idx * 3 + 1], z = means3D[idx * 3 + 2];
    double dmean[3] = {0, 0, 0}, dsc[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 0}, dc6[6] = {0, 0, 0, 0, 0, 0};
    double dcol[3] = {acc[6], acc[7], acc[8]};
    const int vis = g->radius > 0;
    if (vis) {
        const double pvx = v[0] * x + v[4] * y + v[8] * z + v[12];
        const double pvy = v[1] * x + v[5] * y + v[9] * z + v[13];
        const double pvz = v[2] * x + v[6] * y + v[10] * z + v[14];
        const double hx = p[0] * x + p[4] * y + p[8] * z + p[12];
        const double hy = p[1] * x + p[5] * y + p[9] * z + p[13];
        const double hw = p[3] * x + p[7] * y + p[11] * z + p[15];
        const double pw = 1.0 / (hw + 1e-7);
        /* ---- 3D covariance ---- */
        double S[3][3], R[3][3] = {{0}}, sc[3] = {0, 0, 0}, M[3][3] = {{0}};
        double qr = 1, qx = 0, qy = 0, qz = 0;
        if (cov3D) {
            const float* c6 = cov3D + idx * 6;
            S[0][0] = c6[0]; S[0][1] = S[1][0] = c6[1]; S[0][2] = S[2][0] = c6[2];
            S[1][1] = c6[3]; S[1][2] = S[2][1] = c6[4]; S[2][2] = c6[5];
        } else {
            for (int j = 0; j < 3; ++j) sc[j] = (double)s->scale_modifier * scales[idx * 3 + j];
            qr = rotations[idx * 4]; qx = rotations[idx * 4 + 1]; qy = rotations[idx * 4 + 2]; qz = rotations[idx * 4 + 3];
            R[0][0] = 1 - 2 * (qy * qy + qz * qz); R[0][1] = 2 * (qx * qy - qr * qz); R[0][2] = 2 * (qx * qz + qr * qy);
            R[1][0] = 2 * (qx * qy + qr * qz); R[1][1] = 1 - 2 * (qx * qx + qz * qz); R[1][2] = 2 * (qy * qz - qr * qx);
            R[2][0] = 2 * (qx * qz - qr * qy); R[2][1] = 2 * (qy * qz + qr * qx); R[2][2] = 1 - 2 * (qx * qx + qy * qy);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i][j] = R[i][j] * sc[j];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                S[i][j] = 0;
                for (int k = 0; k < 3; ++k) S[i][j] += M[i][k] * M[j][k];
            }
        }
        /* ---- EWA ---- */
        const double fx = (double)W / (2.0 * s->tanfovx), fy = (double)H / (2.0 * s->tanfovy);
        const double limx = 1.3 * (double)s->tanfovx, limy = 1.3 * (double)s->tanfovy;
        const double tz = pvz, txtz = pvx / tz, tytz = pvy / tz;
        const double x_grad_mul = (txtz < -limx || txtz > limx) ? 0.0 : 1.0;      /* upstream: clamped t.x is a constant */
        const double y_grad_mul = (tytz < -limy || tytz > limy) ? 0.0 : 1.0;
        const double tx = fmin(limx, fmax(-limx, txtz)) * tz, ty = fmin(limy, fmax(-limy, tytz)) * tz;
        const double J00 = fx / tz, J02 = -fx * tx / (tz * tz), J11 = fy / tz, J12 = -fy * ty / (tz * tz);
        const double T0[3] = {J00 * v[0] + J02 * v[2], J00 * v[4] + J02 * v[6], J00 * v[8] + J02 * v[10]};
        const double T1[3] = {J11 * v[1] + J12 * v[2], J11 * v[5] + J12 * v[6], J11 * v[9] + J12 * v[10]};
        double ST0[3], ST1[3];
        for (int j = 0; j < 3; ++j) {
            ST0[j] = S[j][0] * T0[0] + S[j][1] * T0[1] + S[j][2] * T0[2];
            ST1[j] = S[j][0] * T1[0] + S[j][1] * T1[1] + S[j][2] * T1[2];
        }
        const double a = T0[0] * ST0[0] + T0[1] * ST0[1] + T0[2] * ST0[2] + LOWPASS;
        const double b = T0[0] * ST1[0] + T0[1] * ST1[1] + T0[2] * ST1[2];
        const double c = T1[0] * ST1[0] + T1[1] * ST1[1] + T1[2] * ST1[2] + LOWPASS;
        const double det = a * c - b * b;
        /* ---- conic (A, B, C) = (c, -b, a) / det; upstream's "denom2inv" = 1 / (det^2 + 1e-7) ---- */
        const double gA = acc[2], gB = acc[3], gC = acc[4];
        const double d2i = 1.0 / (det * det + 1e-7);
        const double da = d2i * (-c * c * gA + b * c * gB + (det - a * c) * gC);
        const double dc = d2i * ((det - a * c) * gA + a * b * gB - a * a * gC);
        const double db = d2i * (2.0 * b * c * gA - (det + 2.0 * b * b) * gB + 2.0 * a * b * gC);
        /* ---- cov2D = T S T^T (+0.3 I):  dL/dS (full matrix), dL/dT ---- */
        double Gm[3][3];
        for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k)
            Gm[j][k] = da * T0[j] * T0[k] + 0.5 * db * (T0[j] * T1[k] + T1[j] * T0[k]) + dc * T1[j] * T1[k];
        double dT0[3], dT1[3];
        for (int j = 0; j < 3; ++j) {
            dT0[j] = 2.0 * da * ST0[j] + db * ST1[j];
            dT1[j] = 2.0 * dc * ST1[j] + db * ST0[j];
        }
        if (cov3D) {     /* packed [xx, xy, xz, yy, yz, zz]: an off-diagonal entry stands for two matrix elements */
            dc6[0] = Gm[0][0]; dc6[1] = 2 * Gm[0][1]; dc6[2] = 2 * Gm[0][2];
            dc6[3] = Gm[1][1]; dc6[4] = 2 * Gm[1][2]; dc6[5] = Gm[2][2];
        } else {         /* S = M M^T, M = R diag(s):  dL/dM = 2 G M */
            double dM[3][3], dR[3][3];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                dM[i][j] = 0;
                for (int k = 0; k < 3; ++k) dM[i][j] += 2.0 * Gm[i][k] * M[k][j];
            }
            for (int j = 0; j < 3; ++j) {
                dsc[j] = (double)s->scale_modifier * (dM[0][j] * R[0][j] + dM[1][j] * R[1][j] + dM[2][j] * R[2][j]);
                for (int i = 0; i < 3; ++i) dR[i][j] = dM[i][j] * sc[j];
            }
            dq[0] = 2 * (-qz * dR[0][1] + qy * dR[0][2] + qz * dR[1][0] - qx * dR[1][2] - qy * dR[2][0] + qx * dR[2][1]);
            dq[1] = 2 * (qy * dR[0][1] + qz * dR[0][2] + qy * dR[1][0] - 2 * qx * dR[1][1] - qr * dR[1][2] + qz * dR[2][0] +
                         qr * dR[2][1] - 2 * qx * dR[2][2]);
            dq[2] = 2 * (-2 * qy * dR[0][0] + qx * dR[0][1] + qr * dR[0][2] + qx * dR[1][0] + qz * dR[1][2] - qr * dR[2][0] +
                         qz * dR[2][1] - 2 * qy * dR[2][2]);
            dq[3] = 2 * (-2 * qz * dR[0][0] - qr * dR[0][1] + qx * dR[0][2] + qr * dR[1][0] - 2 * qz * dR[1][1] + qy * dR[1][2] +
                         qx * dR[2][0] + qy * dR[2][1]);
        }
        /* ---- T = J Rv -> J -> (t.x, t.y, t.z) treated as independent, as upstream does ---- */
        const double dJ00 = dT0[0] * v[0] + dT0[1] * v[4] + dT0[2] * v[8];
        const double dJ02 = dT0[0] * v[2] + dT0[1] * v[6] + dT0[2] * v[10];
        const double dJ11 = dT1[0] * v[1] + dT1[1] * v[5] + dT1[2] * v[9];
        const double dJ12 = dT1[0] * v[2] + dT1[1] * v[6] + dT1[2] * v[10];
        const double tz2 = 1.0 / (tz * tz), tz3 = tz2 / tz;
        const double dtx = x_grad_mul * (-fx * tz2 * dJ02);
        const double dty = y_grad_mul * (-fy * tz2 * dJ12);
        const double dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + 2.0 * fx * tx * tz3 * dJ02 + 2.0 * fy * ty * tz3 * dJ12 +
                           acc[9];                                   /* + dL/d(depth): depth = t.z */
        dmean[0] = dtx * v[0] + dty * v[1] + dtz * v[2];
        dmean[1] = dtx * v[4] + dty * v[5] + dtz * v[6];
        dmean[2] = dtx * v[8] + dty * v[9] + dtz * v[10];
        /* ---- pixel centre -> NDC -> clip -> mean ---- */
        const double dndcx = acc[0] * 0.5 * W, dndcy = acc[1] * 0.5 * H;
        if (dmeans2D) { dmeans2D[idx * 3] = (float)dndcx; dmeans2D[idx * 3 + 1] = (float)dndcy; dmeans2D[idx * 3 + 2] = 0.0f; }
        const double dhx = dndcx * pw, dhy = dndcy * pw, dhw = -(dndcx * hx + dndcy * hy) * pw * pw;
        dmean[0] += dhx * p[0] + dhy * p[1] + dhw * p[3];
        dmean[1] += dhx * p[4] + dhy * p[5] + dhw * p[7];
        dmean[2] += dhx * p[8] + dhy * p[9] + dhw * p[11];
        /* ---- SH colour (module.py:258-266): clamped channels pass no gradient ---- */
        if (shs) {
            const double ux = x - s->campos[0], uy = y - s->campos[1], uz = z - s->campos[2];
            const double n = sqrt(ux * ux + uy * uy + uz * uz);
            const double X = ux / n, Y = uy / n, Z = uz / n;
            const int deg = s->sh_degree, ncoef = (deg + 1) * (deg + 1);
            double basis[16], bp[16], bm[16];
            sh_basis(deg, X, Y, Z, basis);
            double gcol[3];
            for (int ch = 0; ch < 3; ++ch) gcol[ch] = (g->clamped >> ch) & 1u ? 0.0 : dcol[ch];
            const float* sh = shs + (size_t)idx * sh_M * 3;
            /* dL/d(direction) by differentiating the basis polynomials (central differences of a cubic in double with
             * h = 1e-4 are exact to ~1e-8 relative: no hand-copied derivative table to get wrong) */
            double ddir[3] = {0, 0, 0};
            const double h = 1e-4, d0[3] = {X, Y, Z};
            for (int ax = 0; ax < 3; ++ax) {
                double dp[3] = {d0[0], d0[1], d0[2]}, dm[3] = {d0[0], d0[1], d0[2]};
                dp[ax] += h; dm[ax] -= h;
                sh_basis(deg, dp[0], dp[1], dp[2], bp);
                sh_basis(deg, dm[0], dm[1], dm[2], bm);
                for (int k = 0; k < ncoef && k < sh_M; ++k) {
                    const double db_ = (bp[k] - bm[k]) / (2.0 * h);
                    for (int ch = 0; ch < 3; ++ch) ddir[ax] += db_ * sh[k * 3 + ch] * gcol[ch];
                }
            }
            const double dot = X * ddir[0] + Y * ddir[1] + Z * ddir[2];
            dmean[0] += (ddir[0] - X * dot) / n;
            dmean[1] += (ddir[1] - Y * dot) / n;
            dmean[2] += (ddir[2] - Z * dot) / n;
            if (dsh) for (int k = 0; k < sh_M; ++k) for (int ch = 0; ch < 3; ++ch)
                dsh[((size_t)idx * sh_M + k) * 3 + ch] = (float)((k < ncoef && k < 16) ? basis[k] * gcol[ch] : 0.0);
        }
    } else {
        if (dmeans2D) { dmeans2D[idx * 3] = dmeans2D[idx * 3 + 1] = dmeans2D[idx * 3 + 2] = 0.0f; }
        if (shs && dsh) for (int k = 0; k < sh_M * 3; ++k) dsh[(size_t)idx * sh_M * 3 + k] = 0.0f;
    }
    if (dmeans3D) for (int j = 0; j < 3; ++j) dmeans3D[idx * 3 + j] = (float)dmean[j];
    if (dopacity) dopacity[idx] = (float)(vis ? acc[5] : 0.0);
    if (dcolors) for (int j = 0; j < 3; ++j) dcolors[idx * 3 + j] = (float)(vis ? dcol[j] : 0.0);
    if (dscales) for (int j = 0; j < 3; ++j) dscales[idx * 3 + j] = (float)dsc[j];
    if (drot) for (int j = 0; j < 4; ++j) drot[idx * 4 + j] = (float)dq[j];
    if (dcov3D) for (int j = 0; j < 6; ++j) dcov3D[idx * 6 + j] = (float)dc6[j];
}

/*
 * Forward (+ backward when dL_dcolor != NULL).  All pointers are host memory, float32 / int32, contiguous, shapes as the
 * reference passes them (module.py:632-640).  Optional (NULL = skip): shs xor colors_precomp, (scales + rotations) xor
 * cov3D_precomp, out_final_T [H*W], out_n_contrib [H*W] (1-based index of the last blended list entry, upstream's
 * n_contrib), out_margin [H*W] (relative distance of the pixel's closest discrete decision -- alpha >= 1/255, T < 1e-4,
 * power > 0 -- to its threshold, the quantity oracle/raster_oracle.py::ambiguous_pixel_mask thresholds at 1e-4: two fp32
 * implementations may legitimately differ at such a pixel), dL_ddepth, dL_dalpha, every gradient output.  Returns the number of (Gaussian, 16x16 tile) instances
 * (upstream's num_rendered), < 0 on a bad argument / allocation failure.
 */
long exa_oracle_render(const OrcSettings* s, int32_t P, int32_t sh_M, const float* means3D, const float* shs,
                       const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                       const float* cov3D_precomp, float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                       float* out_final_T, int32_t* out_n_contrib, float* out_margin, const float* dL_dcolor, const float* dL_ddepth,
                       const float* dL_dalpha, float* dL_dmeans2D, float* dL_dmeans3D, float* dL_dcolors,
                       float* dL_dopacity, float* dL_dscales, float* dL_drotations, float* dL_dsh, float* dL_dcov3D) {
    if (!s || P < 0 || (P > 0 && (!means3D || !opacities))) return -1;
    if (P > 0 && ((shs == NULL) == (colors_precomp == NULL))) return -1;
    if (P > 0 && (((scales != NULL) != (rotations != NULL)) || ((scales != NULL) == (cov3D_precomp != NULL)))) return -1;
    const int W = s->image_width, H = s->image_height;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, n_tiles = gx * gy;
    Geo* geo = (Geo*)malloc(sizeof(Geo) * (size_t)(P > 0 ? P : 1));
    long* tile_off = (long*)calloc((size_t)n_tiles + 1, sizeof(long));
    if (!geo || !tile_off) { free(geo); free(tile_off); return -2; }

#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        preprocess_one(s, i, means3D, opacities, scales, rotations, cov3D_precomp, gx, gy, &geo[i]);
        if (shs) sh_colour_f32(s, shs + (size_t)i * sh_M * 3, means3D + i * 3, geo[i].col, &geo[i].clamped);
        else { geo[i].col[0] = colors_precomp[i * 3]; geo[i].col[1] = colors_precomp[i * 3 + 1]; geo[i].col[2] = colors_precomp[i * 3 + 2]; }
        if (radii) radii[i] = geo[i].radius;
    }
    /* step 8: per-tile lists, ascending (depth bits, Gaussian index) */
    for (int i = 0; i < P; ++i)
        if (geo[i].radius > 0)
            for (int ty = geo[i].y0; ty < geo[i].y1; ++ty)
                for (int tx = geo[i].x0; tx < geo[i].x1; ++tx) ++tile_off[ty * gx + tx + 1];
    for (int t = 0; t < n_tiles; ++t) tile_off[t + 1] += tile_off[t];
    const long D = tile_off[n_tiles];
    Key* keys = (Key*)malloc(sizeof(Key) * (size_t)(D > 0 ? D : 1));
    long* cur = (long*)malloc(sizeof(long) * (size_t)(n_tiles > 0 ? n_tiles : 1));
    if (!keys || !cur) { free(geo); free(tile_off); free(keys); free(cur); return -2; }
    memcpy(cur, tile_off, sizeof(long) * (size_t)n_tiles);
    for (int i = 0; i < P; ++i)
        if (geo[i].radius > 0) {
            uint32_t bits;
            memcpy(&bits, &geo[i].depth, 4);
            for (int ty = geo[i].y0; ty < geo[i].y1; ++ty)
                for (int tx = geo[i].x0; tx < geo[i].x1; ++tx) {
                    Key* k = &keys[cur[ty * gx + tx]++];
                    k->depth_bits = bits; k->id = i;
                }
        }
    free(cur);
    long max_list = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(max : max_list)
    for (int t = 0; t < n_tiles; ++t) {
        const long n = tile_off[t + 1] - tile_off[t];
        if (n > 1) qsort(keys + tile_off[t], (size_t)n, sizeof(Key), key_cmp);
        if (n > max_list) max_list = n;
    }

    const size_t HW = (size_t)W * H;
    float* final_T = out_final_T ? out_final_T : (float*)malloc(sizeof(float) * (HW ? HW : 1));
    int32_t* n_contrib = out_n_contrib ? out_n_contrib : (int32_t*)malloc(sizeof(int32_t) * (HW ? HW : 1));
    if (!final_T || !n_contrib) return -2;

    /* step 9 / 10: one sequential loop per pixel, like upstream's renderCUDA thread */
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < n_tiles; ++t) {
        const int tx = t % gx, ty = t / gx;
        const Key* list = keys + tile_off[t];
        const long n = tile_off[t + 1] - tile_off[t];
        for (int j = ty * TILE; j < ty * TILE + TILE && j < H; ++j)
            for (int i = tx * TILE; i < tx * TILE + TILE && i < W; ++i) {
                float T = 1.0f, C[3] = {0.f, 0.f, 0.f}, Dp = 0.f, margin = INFINITY;
                int32_t last = 0;
                const float fx = (float)i, fy = (float)j;
                for (long k = 0; k < n; ++k) {
                    const Geo* g = &geo[list[k].id];
                    float dx, dy;
                    const float power = gauss_power(g, fx, fy, &dx, &dy);
                    if (out_margin && fabsf(power) <= 1e-6f) margin = fminf(margin, fabsf(power));
                    if (power > 0.0f) continue;
                    const float alpha = fminf(ALPHA_MAX, g->opacity * expf(power));
                    if (out_margin) margin = fminf(margin, fabsf(alpha - ALPHA_MIN) * 255.0f);
                    if (alpha < ALPHA_MIN) continue;
                    const float test_T = T * (1.0f - alpha);
                    if (out_margin) margin = fminf(margin, fabsf(test_T - T_EPS) / T_EPS);
                    if (test_T < T_EPS) break;                   /* this Gaussian is NOT blended */
                    const float w = alpha * T;
                    C[0] += g->col[0] * w; C[1] += g->col[1] * w; C[2] += g->col[2] * w;
                    Dp += g->depth * w;
                    T = test_T;
                    last = (int32_t)(k + 1);
                }
                const size_t pix = (size_t)j * W + i;
                out_color[pix] = C[0] + T * s->bg[0];
                out_color[HW + pix] = C[1] + T * s->bg[1];
                out_color[2 * HW + pix] = C[2] + T * s->bg[2];
                out_depth[pix] = Dp;
                out_alpha[pix] = 1.0f - T;
                final_T[pix] = T;
                n_contrib[pix] = last;
                if (out_margin) out_margin[pix] = margin;
            }
    }

    if (dL_dcolor) {
        /* per-Gaussian accumulators in double: dpx, dpy, dA, dB, dC, dopacity, dcol[3], ddepth */
        double* acc = (double*)calloc((size_t)(P > 0 ? P : 1) * 10, sizeof(double));
        if (!acc) return -2;
#pragma omp parallel
        {
            float* a_buf = (float*)malloc(sizeof(float) * (size_t)(max_list > 0 ? max_list : 1));
            float* G_buf = (float*)malloc(sizeof(float) * (size_t)(max_list > 0 ? max_list : 1));
#pragma omp for schedule(dynamic, 4)
            for (int t = 0; t < n_tiles; ++t) {
                const int tx = t % gx, ty = t / gx;
                const Key* list = keys + tile_off[t];
                for (int j = ty * TILE; j < ty * TILE + TILE && j < H; ++j)
                    for (int i = tx * TILE; i < tx * TILE + TILE && i < W; ++i) {
                        const size_t pix = (size_t)j * W + i;
                        const int32_t last = n_contrib[pix];
                        if (last == 0) continue;
                        const float fx = (float)i, fy = (float)j;
                        const double g_c[3] = {dL_dcolor[pix], dL_dcolor[HW + pix], dL_dcolor[2 * HW + pix]};
                        const double g_d = dL_ddepth ? dL_ddepth[pix] : 0.0, g_a = dL_dalpha ? dL_dalpha[pix] : 0.0;
                        /* forward decisions again (float32, identical arithmetic): alpha of every blended entry, 0 = skipped */
                        for (int32_t k = 0; k < last; ++k) {
                            const Geo* g = &geo[list[k].id];
                            float dx, dy;
                            const float power = gauss_power(g, fx, fy, &dx, &dy);
                            a_buf[k] = 0.0f; G_buf[k] = 0.0f;
                            if (power > 0.0f) continue;
                            const float Gf = expf(power);
                            const float alpha = fminf(ALPHA_MAX, g->opacity * Gf);
                            if (alpha < ALPHA_MIN) continue;
                            a_buf[k] = alpha; G_buf[k] = Gf;
                        }
                        /* back to front, upstream's recurrence: T_k = T_{k+1} / (1 - alpha_k), suffix colour `rec` */
                        double T = final_T[pix];
                        const double T_final = T;
                        double rec_c[3] = {0, 0, 0}, rec_d = 0, rec_a = 0, last_alpha = 0, last_c[3] = {0, 0, 0}, last_d = 0;
                        const double bg_dot = s->bg[0] * g_c[0] + s->bg[1] * g_c[1] + s->bg[2] * g_c[2];
                        for (int32_t k = last - 1; k >= 0; --k) {
                            const double alpha = a_buf[k];
                            if (alpha == 0.0) continue;
                            const Geo* g = &geo[list[k].id];
                            T = T / (1.0 - alpha);
                            const double w = alpha * T;
                            double dL_dalpha_k = 0.0;
                            for (int ch = 0; ch < 3; ++ch) {
                                rec_c[ch] = last_alpha * last_c[ch] + (1.0 - last_alpha) * rec_c[ch];
                                last_c[ch] = g->col[ch];
                                dL_dalpha_k += (g->col[ch] - rec_c[ch]) * g_c[ch];
                            }
                            rec_d = last_alpha * last_d + (1.0 - last_alpha) * rec_d;
                            last_d = g->depth;
                            dL_dalpha_k += (g->depth - rec_d) * g_d;
                            rec_a = last_alpha * 1.0 + (1.0 - last_alpha) * rec_a;       /* alpha image: "colour" 1 */
                            dL_dalpha_k += (1.0 - rec_a) * g_a;
                            dL_dalpha_k *= T;
                            last_alpha = alpha;
                            dL_dalpha_k += (-T_final / (1.0 - alpha)) * bg_dot;              /* background term */
                            /* min(0.99, .) is straight-through: dalpha/dG = opacity, dalpha/dopacity = G even when clamped */
                            const double Gd = G_buf[k];
                            const double dL_dG = g->opacity * dL_dalpha_k;
                            const double dx = (double)g->px - fx, dy = (double)g->py - fy;
                            const double sG = dL_dG * Gd;                                    /* dL/dpower */
                            double* A = acc + (size_t)list[k].id * 10;
                            const double v0 = sG * -(g->A * dx + g->B * dy), v1 = sG * -(g->C * dy + g->B * dx);
                            const double v2 = sG * -0.5 * dx * dx, v3 = sG * -dx * dy, v4 = sG * -0.5 * dy * dy;
                            const double v5 = Gd * dL_dalpha_k;
#pragma omp atomic
                            A[0] += v0;
#pragma omp atomic
                            A[1] += v1;
#pragma omp atomic
                            A[2] += v2;
#pragma omp atomic
                            A[3] += v3;
#pragma omp atomic
                            A[4] += v4;
#pragma omp atomic
                            A[5] += v5;
                            for (int ch = 0; ch < 3; ++ch) {
                                const double vc = w * g_c[ch];
#pragma omp atomic
                                A[6 + ch] += vc;
                            }
                            const double vd = w * g_d;
#pragma omp atomic
                            A[9] += vd;
                        }
                    }
            }
            free(a_buf); free(G_buf);
        }
#pragma omp parallel for schedule(static)
        for (int i = 0; i < P; ++i)
            preprocess_backward_one(s, i, sh_M, means3D, shs, scales, rotations, cov3D_precomp, &geo[i], acc + (size_t)i * 10,
                                    dL_dmeans2D, dL_dmeans3D, dL_dcolors, dL_dopacity, dL_dscales, dL_drotations, dL_dsh,
                                    dL_dcov3D);
        free(acc);
    }
    if (!out_final_T) free(final_T);
    if (!out_n_contrib) free(n_contrib);
    free(keys); free(tile_off); free(geo);
    return D;
}

int exa_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void exa_oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
