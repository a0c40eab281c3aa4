// Forward per-Gaussian stage for gfx950: cull, EWA projection, conic, radius, tile rect, optional SH
// colour, splat record, exact-footprint sub-tile rect, per-cell histogram (first radix digit).
//
// Replaces upstream `preprocessCUDA` of the third-party rasterizer the reference calls at
// avatar/common/nets/module.py:632-640.  The arithmetic follows
// oracle/raster_oracle.py::preprocess TERM BY TERM and this file is compiled with
// -ffp-contract=off, so px/py/conic/radius/rect are bit-identical to the float32 oracle.
//
// HBM traffic per Gaussian: reads mean 12 + scale 12 + quat 16 + opacity 4 + colour 12 = 56 B,
// writes radius 4 + one 64-byte splat record (one line, later gathered whole by the per-pixel kernels).
#include "common.h"

namespace exa {

__device__ __forceinline__ float sh_channel(int deg, const float* __restrict__ sh, int c,
                                            float x, float y, float z) {
    // reference avatar/common/utils/transforms.py:112-167 (polynomials), coefficients [M][3]
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
    const float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                C3_6 = -0.5900435899266435f;
    float res = C0 * sh[0 * 3 + c];
    if (deg > 0) {
        res = res - C1 * y * sh[1 * 3 + c] + C1 * z * sh[2 * 3 + c] - C1 * x * sh[3 * 3 + c];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = res + C2_0 * xy * sh[4 * 3 + c] + C2_1 * yz * sh[5 * 3 + c] +
                  C2_2 * (2.0f * zz - xx - yy) * sh[6 * 3 + c] + C2_3 * xz * sh[7 * 3 + c] +
                  C2_4 * (xx - yy) * sh[8 * 3 + c];
            if (deg > 2) {
                res = res + C3_0 * y * (3.0f * xx - yy) * sh[9 * 3 + c] + C3_1 * xy * z * sh[10 * 3 + c] +
                      C3_2 * y * (4.0f * zz - xx - yy) * sh[11 * 3 + c] +
                      C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * 3 + c] +
                      C3_4 * x * (4.0f * zz - xx - yy) * sh[13 * 3 + c] + C3_5 * z * (xx - yy) * sh[14 * 3 + c] +
                      C3_6 * x * (xx - 3.0f * yy) * sh[15 * 3 + c];
            }
        }
    }
    return res;
}

__device__ __forceinline__ int floor_div8(int v) { return v >= 0 ? (v >> 3) : -((-v + 7) >> 3); }

// One Gaussian: oracle steps 1-7 + splat record.  Returns the number of sub-tile instances and the
// sub-tile rect through sxy (sx0, sx1, sy0, sy1; upper bounds exclusive).
struct ShColour { float r, g, b; uint32_t flags; };           // SH colour evaluated from the LDS-staged coefficients

// Colour of one Gaussian from its SH coefficients (reference module.py:258-266 semantics): direction from the camera
// centre, + 0.5, clamped at 0 with the clamp recorded per channel for the backward.
__device__ __forceinline__ ShColour sh_colour(const PreprocessArgs& a, const float* sh, float x, float y, float z) {
    const float* cp = a.campos;
    float dx = x - cp[0], dy = y - cp[1], dz = z - cp[2];
    const float inv = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
    dx *= inv; dy *= inv; dz *= inv;
    ShColour c;
    c.flags = 0u;
    c.r = sh_channel(a.sh_degree, sh, 0, dx, dy, dz) + 0.5f;
    c.g = sh_channel(a.sh_degree, sh, 1, dx, dy, dz) + 0.5f;
    c.b = sh_channel(a.sh_degree, sh, 2, dx, dy, dz) + 0.5f;
    if (c.r < 0.f) { c.r = 0.f; c.flags |= 1u; }
    if (c.g < 0.f) { c.g = 0.f; c.flags |= 2u; }
    if (c.b < 0.f) { c.b = 0.f; c.flags |= 4u; }
    return c;
}

__device__ __forceinline__ uint32_t preprocess_one(const PreprocessArgs& a, int idx, int sxy[4], bool& visible,
                                                   const ShColour& shc, uint32_t& tiles16) {
    tiles16 = 0u;
    const float* __restrict__ v = a.viewmatrix;
    const float* __restrict__ p = a.projmatrix;
    const float x = a.means3D[idx * 3 + 0], y = a.means3D[idx * 3 + 1], z = a.means3D[idx * 3 + 2];
    // issue every input load up front (one memory round trip instead of a dependent chain)
    const float op = a.opacities[idx];
    float in_s0 = 0.f, in_s1 = 0.f, in_s2 = 0.f, in_c0 = 0.f, in_c1 = 0.f, in_c2 = 0.f;
    float4 in_q = make_float4(1.f, 0.f, 0.f, 0.f);
    float in_cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.cov3D_precomp) {
#pragma unroll
        for (int i = 0; i < 6; ++i) in_cov[i] = a.cov3D_precomp[idx * 6 + i];
    } else {
        in_s0 = a.scales[idx * 3 + 0]; in_s1 = a.scales[idx * 3 + 1]; in_s2 = a.scales[idx * 3 + 2];
        in_q = reinterpret_cast<const float4*>(a.rotations)[idx];
    }
    if (!a.shs) {
        in_c0 = a.colors_precomp[idx * 3 + 0]; in_c1 = a.colors_precomp[idx * 3 + 1]; in_c2 = a.colors_precomp[idx * 3 + 2];
    }

    // 1. view space
    const float pvx = ((v[0] * x + v[4] * y) + v[8] * z) + v[12];
    const float pvy = ((v[1] * x + v[5] * y) + v[9] * z) + v[13];
    const float pvz = ((v[2] * x + v[6] * y) + v[10] * z) + v[14];

    uint4* rec = reinterpret_cast<uint4*>(a.splats + idx);
    visible = pvz > NEAR_CULL;
    float px = 0.f, py = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, ca2 = 0.f, cc2 = 0.f;
    int radius = 0;
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    if (visible) {
        // 2. clip space, perspective divide
        const float hx = ((p[0] * x + p[4] * y) + p[8] * z) + p[12];
        const float hy = ((p[1] * x + p[5] * y) + p[9] * z) + p[13];
        const float hw = ((p[3] * x + p[7] * y) + p[11] * z) + p[15];
        const float pw = 1.0f / (hw + 1e-7f);
        const float ndcx = hx * pw, ndcy = hy * pw;
        // 3. 3D covariance
        float S00, S01, S02, S11, S12, S22;
        if (a.cov3D_precomp) {
            S00 = in_cov[0]; S01 = in_cov[1]; S02 = in_cov[2]; S11 = in_cov[3]; S12 = in_cov[4]; S22 = in_cov[5];
        } else {
            const float s0 = a.scale_modifier * in_s0;
            const float s1 = a.scale_modifier * in_s1;
            const float s2 = a.scale_modifier * in_s2;
            const float4 q = in_q;
            const float qr = q.x, qx = q.y, qy = q.z, qz = q.w;
            const float R00 = 1.0f - 2.0f * (qy * qy + qz * qz), R01 = 2.0f * (qx * qy - qr * qz),
                        R02 = 2.0f * (qx * qz + qr * qy);
            const float R10 = 2.0f * (qx * qy + qr * qz), R11 = 1.0f - 2.0f * (qx * qx + qz * qz),
                        R12 = 2.0f * (qy * qz - qr * qx);
            const float R20 = 2.0f * (qx * qz - qr * qy), R21 = 2.0f * (qy * qz + qr * qx),
                        R22 = 1.0f - 2.0f * (qx * qx + qy * qy);
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
        // 4. EWA projection
        const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
        const float tz = pvz;
        const float tx = fminf(limx, fmaxf(-limx, pvx / tz)) * tz;
        const float ty = fminf(limy, fmaxf(-limy, pvy / tz)) * tz;
        const float J00 = a.focal_x / tz, J02 = -(a.focal_x * tx) / (tz * tz);
        const float J11 = a.focal_y / tz, J12 = -(a.focal_y * ty) / (tz * tz);
        const float T00 = J00 * v[0] + J02 * v[2], T01 = J00 * v[4] + J02 * v[6], T02 = J00 * v[8] + J02 * v[10];
        const float T10 = J11 * v[1] + J12 * v[2], T11 = J11 * v[5] + J12 * v[6], T12 = J11 * v[9] + J12 * v[10];
        const float U00 = (T00 * S00 + T01 * S01) + T02 * S02;
        const float U01 = (T00 * S01 + T01 * S11) + T02 * S12;
        const float U02 = (T00 * S02 + T01 * S12) + T02 * S22;
        const float U10 = (T10 * S00 + T11 * S01) + T12 * S02;
        const float U11 = (T10 * S01 + T11 * S11) + T12 * S12;
        const float U12 = (T10 * S02 + T11 * S12) + T12 * S22;
        ca2 = ((U00 * T00 + U01 * T01) + U02 * T02) + LOWPASS;
        const float cb2 = (U00 * T10 + U01 * T11) + U02 * T12;
        cc2 = ((U10 * T10 + U11 * T11) + U12 * T12) + LOWPASS;
        const float det = ca2 * cc2 - cb2 * cb2;
        if (det == 0.0f) {
            visible = false;
        } else {
            const float det_inv = 1.0f / det;
            ca = cc2 * det_inv; cb = -cb2 * det_inv; cc = ca2 * det_inv;
            // 5. radius
            const float mid = 0.5f * (ca2 + cc2);
            const float lam = mid + sqrtf(fmaxf(mid * mid - det, 0.1f));
            const float rf = ceilf(3.0f * sqrtf(lam));
            // 6. pixel centre
            px = ((ndcx + 1.0f) * a.grid.W - 1.0f) * 0.5f;
            py = ((ndcy + 1.0f) * a.grid.H - 1.0f) * 0.5f;
            // 7. tile rect (C float->int truncation, clamped to the grid)
            const float big = 1048576.0f;
            auto tr = [&](float t) -> int {
                t = (t != t) ? 0.0f : fminf(fmaxf(t, -big), big);
                return (int)t;
            };
            x0 = min(a.grid.gx, max(0, tr((px - rf) / TILE)));
            x1 = min(a.grid.gx, max(0, tr(((px + rf) + (TILE - 1)) / TILE)));
            y0 = min(a.grid.gy, max(0, tr((py - rf) / TILE)));
            y1 = min(a.grid.gy, max(0, tr(((py + rf) + (TILE - 1)) / TILE)));
            radius = (int)fminf(rf, 2147483520.0f);
            if ((x1 - x0) * (y1 - y0) == 0) visible = false;
        }
    }
    sxy[0] = sxy[1] = sxy[2] = sxy[3] = 0;
    if (!visible) {
        a.radii[idx] = 0;
        if (a.is_vis) a.is_vis[idx] = 0;
        const uint4 zero = make_uint4(0, 0, 0, 0);
        rec[0] = zero; rec[1] = zero; rec[2] = zero; rec[3] = zero;
        return 0;
    }
    a.radii[idx] = radius;
    if (a.is_vis) a.is_vis[idx] = radius > 0;
    tiles16 = (uint32_t)((x1 - x0) * (y1 - y0));               // upstream's tiles_touched
    // colour: precomputed, or SH evaluated here (reference module.py:258-266 semantics)
    float cr, cg, cbl;
    uint32_t flags = 0;
    if (a.shs) {
        cr = shc.r; cg = shc.g; cbl = shc.b; flags = shc.flags;
    } else {
        cr = in_c0;
        cg = in_c1;
        cbl = in_c2;
    }
    // Sub-tile rect: inside the upstream 16x16-tile rect AND intersecting the exact bounding box of
    // {alpha >= 1/255}: 0.5 d^T Q d <= tau = ln(255 o), whose half extents are sqrt(2 tau cov_xx / yy).
    // (Conservative: +0.1 % and +0.01 px; a sub-tile outside it holds no pixel that passes the
    // per-pixel alpha test, so dropping it cannot change the image.)
    uint32_t n_inst = 0;
    const float o255 = 255.0f * op;
    if (o255 >= 1.0f) {
        const float tau2 = 2.0f * __logf(o255) + 1e-3f;
        const float ex = sqrtf(tau2 * ca2) * 1.001f + 0.01f;
        const float ey = sqrtf(tau2 * cc2) * 1.001f + 0.01f;
        const float lim = 1.0e6f;
        const int bx0 = (int)floorf(fminf(fmaxf(px - ex, -lim), lim)), bx1 = (int)ceilf(fminf(fmaxf(px + ex, -lim), lim));
        const int by0 = (int)floorf(fminf(fmaxf(py - ey, -lim), lim)), by1 = (int)ceilf(fminf(fmaxf(py + ey, -lim), lim));
        sxy[0] = max(2 * x0, floor_div8(bx0));
        sxy[1] = min(min(2 * x1, a.grid.sx), floor_div8(bx1) + 1);
        sxy[2] = max(2 * y0, floor_div8(by0));
        sxy[3] = min(min(2 * y1, a.grid.sy), floor_div8(by1) + 1);
        if (sxy[1] > sxy[0] && sxy[3] > sxy[2]) n_inst = (uint32_t)(sxy[1] - sxy[0]) * (uint32_t)(sxy[3] - sxy[2]);
        else sxy[0] = sxy[1] = sxy[2] = sxy[3] = 0;
    }
    rec[0] = make_uint4(__float_as_uint(px), __float_as_uint(py), flags, (uint32_t)radius);
    // conic pre-scaled for the per-pixel kernels: log2 G = A' dx^2 + B' dx dy + C' dy^2 (blend.h)
    rec[1] = make_uint4(__float_as_uint((-0.5f * LOG2E) * ca), __float_as_uint(-LOG2E * cb), __float_as_uint((-0.5f * LOG2E) * cc),
                        __float_as_uint(op));
    rec[2] = make_uint4(__float_as_uint(cr), __float_as_uint(cg), __float_as_uint(cbl), __float_as_uint(pvz));
    rec[3] = make_uint4((uint32_t)sxy[0] | ((uint32_t)sxy[1] << 16), (uint32_t)sxy[2] | ((uint32_t)sxy[3] << 16),
                        n_inst, 0u);
    return n_inst;
}

// CHUNK Gaussians per workgroup.  Besides the per-Gaussian work, builds this chunk's histogram over the
// 64x64-pixel cells in LDS (entries and instances packed in one u64) and stores it as one row of the
// (chunk, cell) count matrix -- no global atomics at all: device-scope atomics run at only ~12 G/s chip-wide on
// MI355X and ~150 chunks adding into the same few dozen cell counters serialise at the memory side
// (that tail was ~8 us of this kernel).  cell_scan_kernel sums the columns.
constexpr int PBLOCK_PLAIN = 1024;     // threads per chunk: one Gaussian each (no atomics left to contend on)
constexpr int PBLOCK_SH = 256;         // SH variant: four Gaussians per thread, one after the other; all four waves stage at once
// SH coefficients ([P][M][3] floats, up to 192 B per Gaussian) are staged through LDS: a lane reading its own 48 floats
// straight from global memory touches 64 different cache lines per load instruction, 16 waves of that thrash the L1 and
// every dword load goes to L2 (C5, 300 k Gaussians at degree 3: 65 us for 75 MB).  Instead each wave reads the
// CONTIGUOUS block of its 64 Gaussians with coalesced dword loads (all issued up front, held in registers), and
// the waves take turns -- four at a time, 4 x 12.25 KiB -- to transpose their block through LDS (row stride F | 1 words:
// conflict-free both ways) and evaluate the colours.
constexpr int SH_MAX_F = 48;           // floats per Gaussian at degree 3
constexpr int SH_WAVES = 4;            // waves staging at a time
constexpr int SH_LDS_FLOATS = SH_WAVES * 64 * (SH_MAX_F | 1);
// One wave: the contiguous block of its 64 Gaussians' coefficients (nf floats of it exist), global -> LDS rows of F | 1
// words.  The whole block is in flight at once (48 KiB per phase and workgroup: enough bytes in flight to stream at HBM
// rate); full blocks are loaded unconditionally (predicated loads compiled to 48 exec-mask branches), and the scheduling
// barrier keeps the LDS addresses from being computed -- and held in registers -- while the loads are still out.
template <int F>
__device__ __forceinline__ void stage_sh_block(const float* __restrict__ blk, int nf, float* s_sh, int lane) {
    constexpr int S = F | 1, DQ = 64 / F, DR = 64 % F;
    int l2 = lane;                                              // opaque copy: the LDS addresses below depend on the lane only,
    asm volatile("" : "+v"(l2));                                // and hoisted to the kernel's prologue they cost 200 registers
    int g = l2 / F, j = l2 % F;                                 // float i * 64 + lane is coefficient j of Gaussian g of the block
    if (nf < 64 * F) {                                          // the array's last, partial block: one float at a time
#pragma unroll 1
        for (int i = 0; i < F; ++i) {
            if (i * 64 + lane < nf) s_sh[g * S + j] = blk[i * 64 + lane];
            g += DQ; j += DR;
            if (j >= F) { j -= F; g += 1; }
        }
        return;
    }
    float v[F];
#pragma unroll
    for (int i = 0; i < F; ++i) v[i] = blk[i * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < F; ++i) {
        s_sh[g * S + j] = v[i];
        g += DQ; j += DR;
        if (j >= F) { j -= F; g += 1; }
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);    // (else all F addresses are computed first: +48 registers)
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <bool SH, int PBLOCK>
__global__ __launch_bounds__(PBLOCK) void preprocess_fwd_kernel(Batch<PreprocessArgs> batch) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_cell[];   // [cells] (+ SH staging area behind it)
    __shared__ uint32_t s_red[PBLOCK / 64];
    const PreprocessArgs& a = batch.v[blockIdx.y];              // this workgroup's job (kernarg segment: scalar loads)
    if ((int)blockIdx.x >= num_chunks(a.P)) return;             // a job with fewer Gaussians than the largest of the batch
    const int tid = threadIdx.x;
#ifdef EXA_PROBE_PFWD      // probe build only (tools/gpu_pfwd_phases.py): phases of every workgroup, 100 MHz clock, into the radii
                           // of the chunk's first Gaussians (which the probe run does not use)
    const unsigned long long pf_t0 = wall_clock64();
#define PFWD_PHASE(i) do { __syncthreads(); if (tid == 0) a.radii[blockIdx.x * CHUNK + (i)] = (int)(wall_clock64() - pf_t0); } while (0)
#else
#define PFWD_PHASE(i) do { } while (0)
#endif
    for (int c = tid; c < a.grid.cells; c += PBLOCK) s_cell[c] = 0ull;
    __syncthreads();
    uint32_t inst_sum = 0;
    uint32_t nvis = 0, ntiles = 0;
#pragma unroll 1
    for (int it = 0; it < CHUNK / PBLOCK; ++it) {
        const int idx = blockIdx.x * CHUNK + it * PBLOCK + tid;
        ShColour shc = {0.f, 0.f, 0.f, 0u};
        if (SH && a.shs) {                                      // workgroup-uniform
            const int F = a.sh_M * 3, S = F | 1;                // floats per Gaussian, LDS row stride
            const int lane = tid & 63, wave = tid >> 6;
            const int g0 = idx - lane;                          // first Gaussian of this wave
            const float* blk = a.shs + (size_t)g0 * F;
            const int nf = min(64, max(0, a.P - g0)) * F;       // floats of the block that exist
            float mx = 0.f, my = 0.f, mz = 0.f;
            if (idx < a.P) { mx = a.means3D[idx * 3 + 0]; my = a.means3D[idx * 3 + 1]; mz = a.means3D[idx * 3 + 2]; }
            float* s_sh = reinterpret_cast<float*>(s_cell + a.grid.cells) + (wave % SH_WAVES) * 64 * (SH_MAX_F | 1);
#pragma unroll 1
            for (int ph = 0; ph < PBLOCK / 64 / SH_WAVES; ++ph) {
                if (wave / SH_WAVES == ph) {
                    if (nf > 0) {
                        if (F == 48) stage_sh_block<48>(blk, nf, s_sh, lane);
                        else if (F == 27) stage_sh_block<27>(blk, nf, s_sh, lane);
                        else if (F == 12) stage_sh_block<12>(blk, nf, s_sh, lane);
                        else stage_sh_block<3>(blk, nf, s_sh, lane);
                    }
                    wave_lds_fence();
                    if (idx < a.P) shc = sh_colour(a, s_sh + lane * S, mx, my, mz);
                }
                __syncthreads();
            }
        }
        if (!SH && a.shs && idx < a.P)                          // image too large for the staging area: straight from global
            shc = sh_colour(a, a.shs + (size_t)idx * a.sh_M * 3, a.means3D[idx * 3 + 0], a.means3D[idx * 3 + 1], a.means3D[idx * 3 + 2]);
        if (idx < a.P) {
            int sxy[4];
            bool vis;
            uint32_t t16;
            const uint32_t n = preprocess_one(a, idx, sxy, vis, shc, t16);
            nvis += vis ? 1u : 0u;
            ntiles += t16;
            if (n) {
                inst_sum += n;
                const int cx0 = sxy[0] >> 3, cx1 = (sxy[1] - 1) >> 3, cy0 = sxy[2] >> 3, cy1 = (sxy[3] - 1) >> 3;
                for (int cy = cy0; cy <= cy1; ++cy) {
                    const int hy = min(sxy[3], (cy + 1) * CELL_SUBS) - max(sxy[2], cy * CELL_SUBS);
                    for (int cx = cx0; cx <= cx1; ++cx) {
                        const int hx = min(sxy[1], (cx + 1) * CELL_SUBS) - max(sxy[0], cx * CELL_SUBS);
                        const unsigned long long add = ((unsigned long long)(uint32_t)(hx * hy) << 32) | 1ull;
                        __hip_atomic_fetch_add(&s_cell[cy * a.grid.cx + cx], add, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
    }
    PFWD_PHASE(1);                                              // Gaussians projected, records written, LDS histogram
    // chunk totals
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        inst_sum += __shfl_xor(inst_sum, d, 64);
        nvis += __shfl_xor(nvis, d, 64);
        ntiles += __shfl_xor(ntiles, d, 64);
    }
    __shared__ uint32_t s_vis[PBLOCK / 64], s_til[PBLOCK / 64];
    if ((tid & 63) == 0) { s_red[tid >> 6] = inst_sum; s_vis[tid >> 6] = nvis; s_til[tid >> 6] = ntiles; }
    __syncthreads();
    if (tid == 0) {
        { uint32_t t = 0; for (int i = 0; i < PBLOCK / 64; ++i) t += s_red[i]; a.tw.chunk_inst[blockIdx.x] = t; }
        { uint32_t t = 0; for (int i = 0; i < PBLOCK / 64; ++i) t += s_vis[i]; a.tw.chunk_vis[blockIdx.x] = t; }
        { uint32_t t = 0; for (int i = 0; i < PBLOCK / 64; ++i) t += s_til[i]; a.tw.chunk_tiles[blockIdx.x] = t; }
    }
    // this chunk's row of the (chunk, cell) count matrix: plain coalesced stores, zeros included
    unsigned long long* row = a.tw.chunk_cell + (size_t)blockIdx.x * a.grid.cells;
    for (int c = tid; c < a.grid.cells; c += PBLOCK) row[c] = s_cell[c];
    PFWD_PHASE(2);                                              // chunk totals + matrix row stored
#ifdef EXA_PROBE_PFWD
    if (tid == 0) a.radii[blockIdx.x * CHUNK + 3] = (int)(pf_t0 & 0xffffff);
#endif
}

__global__ __launch_bounds__(BLOCK) void mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                             const float* __restrict__ v, uint8_t* present) {
    const int idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= P) return;
    const float x = means3D[idx * 3 + 0], y = means3D[idx * 3 + 1], z = means3D[idx * 3 + 2];
    const float pvz = ((v[2] * x + v[6] * y) + v[10] * z) + v[14];
    present[idx] = pvz > NEAR_CULL ? 1 : 0;
}

// Camera block of GaussianRenderer.forward (reference avatar/common/nets/module.py:604-608) from DEVICE-resident R, t:
//   viewmatrix = [[R, t], [0, 0, 0, 1]]^T,  projmatrix = viewmatrix @ proj^T,  campos = -R^T t (= inverse(viewmatrix)[3, :3])
// One 64-lane launch instead of ~25 tiny tensor ops (or a read-back + host math + upload) per new camera.  `proj` is
// get_proj_matrix's [4, 4] (transforms.py:43-64; depends on focal and image size only), row-major, passed by value.
// The sums run k = 0..3 left to right (this file is compiled with -ffp-contract=off); the columns the kernels read
// (0, 1, 3) have a single non-zero term each, so they equal what any matmul computes.
// Optional focal-length check: `proj` (and the tan(fov) baked into the rasterizer's kernel arguments) were derived on the
// host from a focal length the caller REMEMBERS; when `focal` (device float[2]) is given, lane 19 compares it with that
// memory (fx, fy) and reports {equal, 0, 0, tag} into 16 bytes of pinned host memory (same zero-copy scheme as the header
// report in binning.hip), so a caller that gets a fresh focal tensor every frame never has to read it back.
typedef uint32_t cam_v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void camera_block_kernel(const float* __restrict__ R, const float* __restrict__ t, Proj16 proj,
                                                          float* __restrict__ view_out, float* __restrict__ proj_out,
                                                          float* __restrict__ campos_out, const float* __restrict__ focal,
                                                          float fx, float fy, uint32_t* host_flag, uint32_t tag) {
    const int l = threadIdx.x;
    if (l == 19 && focal && host_flag) {
        const cam_v4u v = {(focal[0] == fx && focal[1] == fy) ? 1u : 0u, 0u, 0u, tag};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(host_flag), "v"(v) : "memory");
    }
    if (l < 16) {
        const int i = l >> 2, j = l & 3;                        // row i, column j of the TRANSPOSED view matrix
        auto vt = [&](int r, int c) -> float {                  // view^T[r][c] = view[c][r]
            return c < 3 ? (r < 3 ? R[c * 3 + r] : t[c]) : (r < 3 ? 0.0f : 1.0f);
        };
        view_out[l] = vt(i, j);
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = acc + vt(i, k) * proj.m[j * 4 + k];     // proj^T[k][j] = proj[j][k]
        proj_out[l] = acc;
    } else if (l < 19) {
        const int j = l - 16;
        campos_out[j] = -((R[0 * 3 + j] * t[0] + R[1 * 3 + j] * t[1]) + R[2 * 3 + j] * t[2]);
    }
}

hipError_t launch_camera_block(const float* R, const float* t, const Proj16& proj, float* view_out, float* proj_out,
                               float* campos_out, const float* focal, float fx, float fy, uint32_t* host_flag, uint32_t tag,
                               hipStream_t s) {
    camera_block_kernel<<<1, 64, 0, s>>>(R, t, proj, view_out, proj_out, campos_out, focal, fx, fy, host_flag, tag);
    return hipGetLastError();
}

hipError_t launch_preprocess_fwd(const PreprocessArgs* a, int K, hipStream_t s) {
    Batch<PreprocessArgs> b;
    int chunks = 0, cells = 0;
    for (int k = 0; k < K; ++k) {
        b.v[k] = a[k];
        chunks = max(chunks, num_chunks(a[k].P));
        cells = max(cells, a[k].grid.cells);
    }
    for (int k = K; k < MAX_BATCH; ++k) b.v[k] = a[0];
    if (chunks == 0) return hipSuccess;
    bool sh = false;
    for (int k = 0; k < K; ++k) sh = sh || a[k].shs != nullptr;
    sh = sh && (size_t)cells * 8 + SH_LDS_FLOATS * sizeof(float) <= 64 * 1024;     // default dynamic-LDS limit
    if (sh) preprocess_fwd_kernel<true, PBLOCK_SH><<<dim3(chunks, K), PBLOCK_SH, (size_t)cells * 8 + SH_LDS_FLOATS * sizeof(float), s>>>(b);
    else preprocess_fwd_kernel<false, PBLOCK_PLAIN><<<dim3(chunks, K), PBLOCK_PLAIN, (size_t)cells * 8, s>>>(b);
    return hipGetLastError();
}

hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present,
                               hipStream_t s) {
    if (P == 0) return hipSuccess;
    mark_visible_kernel<<<(P + BLOCK - 1) / BLOCK, BLOCK, 0, s>>>(P, means3D, viewmatrix, present);
    return hipGetLastError();
}

}  // namespace exa
