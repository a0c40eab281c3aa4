// Per-batch back-to-front backward of the alpha compositing for gfx950 -- atomic-free, one wave per batch.
//
// A single wave issues roughly one VALU instruction per 5 cycles on gfx950, so a kernel that walks a
// sub-tile's whole list with one wave is bound by its longest list (measured: 177 us, waves idle 70 % of
// their life).  Here the unit of work is a BATCH SLOT: 64 consecutive entries of one sub-tile's sorted list
// and the 64 pixels of that sub-tile (lane l -> pixel (l & 7, l >> 3)).  The forward pass checkpoints the
// per-pixel state (T, C_rgb, depth) at the start of every batch and at its exit; a batch restarts the
// back-to-front recurrence from the state at its own end:
//     T_e, C_e = ckpt[slot + 1],   C_fin = ckpt[exit slot],   rec_e = (C_fin - C_e) / T_e   (normalised suffix)
// so every batch of every sub-tile runs concurrently and the longest chain is 64 splats.
// For every splat with at least one contributing lane the ten partial sums are reduced across the wave with
// a packed DPP / permlane-swap butterfly (two transposing quad_perm steps shrink 10 registers to 3, then
// row_shr:4/8 + v_permlane16/32_swap -- no LDS traffic) and four lanes park them in LDS; at the end the 64
// lanes store their splat's 48-byte `Partial` to its Gaussian-major slot with plain stores.  Every instance
// gets its slot written exactly once (zeros when nothing contributed): no memset, NO atomic in the whole
// backward pass (device-scope fp32 atomics run at ~12 G/s on MI355X).
//
// Replaces upstream BACKWARD::renderCUDA of the rasterizer behind reference
// avatar/common/nets/module.py:632-640 (backward reached from avatar/main/train.py:46).
// Derivatives follow oracle/raster_oracle.py (autograd of oracle step 9/10) with straight-through
// min(0.99, .).  The screen-space gradients are emitted as five moments of s = dL/dG * G
// (sum s dx, s dy, s dx^2, s dx dy, s dy^2); preprocess_bwd.hip turns them into d/d(mean2D, conic).
//
// Algorithmic HBM bytes: reads 4 B/instance (sorted ids), 64 B per gathered splat, 12 (+8) B/pixel of
// incoming gradient and 8 B/pixel (final_T, n_contrib) per batch, 2 x 20 B/pixel of checkpoints per batch;
// writes 48 B per instance.
#include "common.h"

namespace exa {

constexpr int RBLOCK = 64;            // ONE wave per workgroup
constexpr int NACC = 10;              // mx my mxx mxy myy dop dr dg db dz

// ---- packed wave64 reduction of the ten partial sums ------------------------------------------------
// A plain DPP reduction costs 6 steps x 10 values.  Here the first two butterfly steps (lane ^ 1, lane ^ 2,
// DPP quad_perm) also TRANSPOSE: a lane keeps half of its registers and ships the other half, so 10
// registers shrink to 5 and then to 3, each holding four different sums selected by (lane & 3).  The
// remaining steps (row_shr:4, row_shr:8, v_permlane16_swap, v_permlane32_swap) only run on 3 registers.
// Result: lanes 12..15 of every row hold   q0 = {v0,v1,v2,v3}[lane&3], q1 = {v4..v7}, q2 = {v8,v9,v8,v9}.
__device__ __forceinline__ float dpp_quad_xor1(float x) {      // quad_perm [1,0,3,2]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_quad_xor2(float x) {      // quad_perm [2,3,0,1]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
}
__device__ __forceinline__ float reduce_rows_and_wave(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, true));  // row_shr:4
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, true));  // row_shr:8
    // v_permlane{16,32}_swap exchange halves BETWEEN two registers (vdst odd rows / upper half <-> src even
    // rows / lower half).  Inline asm: with hipcc 7.2 the builtins return the new vdst in BOTH result slots
    // (probed on gfx950, tools/probe/reduce_probe.hip), so the second register would be lost.
    // `s_nop 1` = the two wait states a VALU write needs before v_permlane*_swap reads it.
    {   // rows 0<->1, 2<->3:  a = [x0 x0 x2 x2], b = [x1 x1 x3 x3]
        float a = x, b = x;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        x = a + b;
    }
    {   // halves:  a = [lo lo], b = [hi hi]
        float a = x, b = x;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        x = a + b;
    }
    return x;
}
__device__ __forceinline__ void packed_reduce10(const float (&v)[10], bool odd, bool hi, float& q0, float& q1, float& q2) {
    float r[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        const float keep = odd ? b : a, send = odd ? a : b;
        r[i] = keep + dpp_quad_xor1(send);                 // even lanes: pair-sum of a, odd lanes: pair-sum of b
    }
    {
        const float keep = hi ? r[1] : r[0], send = hi ? r[0] : r[1];
        q0 = keep + dpp_quad_xor2(send);                   // lane&3 -> quad sums of v0, v1, v2, v3
    }
    {
        const float keep = hi ? r[3] : r[2], send = hi ? r[2] : r[3];
        q1 = keep + dpp_quad_xor2(send);                   // v4 .. v7
    }
    q2 = r[4] + dpp_quad_xor2(r[4]);                       // v8, v9, v8, v9
    q0 = reduce_rows_and_wave(q0);
    q1 = reduce_rows_and_wave(q1);
    q2 = reduce_rows_and_wave(q2);
}

__global__ __launch_bounds__(RBLOCK) void render_bwd_kernel(RenderBwdArgs a) {
    __shared__ float4 s_g0[64];
    __shared__ float4 s_g1[64];
    __shared__ float4 s_g2[64];
    __shared__ float4 s_out[64][3];             // the Partial of each staged splat

    const int lane = threadIdx.x;
    const uint32_t own = a.bw.owner[blockIdx.x];
    if (own == 0) return;                                       // unused batch slot
    const int st = (int)own - 1;
    const uint2 range = a.tw.ranges[st];
    const int n = (int)(range.y - range.x);
    const int b0 = (int)(range.x / BATCH);                      // first slot of the sub-tile
    const int bq = (int)blockIdx.x - b0;                        // batch index inside the sub-tile
    const int bstart = bq * BATCH;
    const int cnt = min(BATCH, n - bstart);
    const uint2 fe = a.tw.fwd_exit[st];
    const int n_eff = (int)fe.x;                                // last list position any pixel blended
    const SubTile sub = decode_subtile(st, a.grid);

    // stage this batch: ids -> records; every lane also computes its splat's Partial slot
    const Splat* __restrict__ splats = a.splats;
    uint32_t pslot = 0;
    if (lane < cnt) {
        const uint32_t id = a.bw.sorted[range.x + bstart + lane];
        const float4* rec = reinterpret_cast<const float4*>(splats + id);
        const uint4 r3 = reinterpret_cast<const uint4*>(rec)[3];
        const int sx0 = r3.x & 0xffff, sx1 = r3.x >> 16, sy0 = r3.y & 0xffff;
        pslot = r3.w + (uint32_t)((sub.gsy - sy0) * (sx1 - sx0) + (sub.gsx - sx0));
        s_g0[lane] = rec[0];
        s_g1[lane] = rec[1];
        s_g2[lane] = rec[2];
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    s_out[lane][0] = zero4; s_out[lane][1] = zero4; s_out[lane][2] = zero4;

    if (bstart < n_eff) {
        const int pxi = sub.ox + (lane & 7), pyi = sub.oy + (lane >> 3);
        const bool inside = pxi < a.grid.W && pyi < a.grid.H;
        const float fx = (float)pxi, fy = (float)pyi;
        const size_t HW = (size_t)a.grid.W * a.grid.H;
        const size_t pix = (size_t)pyi * a.grid.W + pxi;
        float gr = 0.f, gg = 0.f, gb = 0.f, gd = 0.f, ga = 0.f, T_final = 1.f;
        int last = 0;
        if (inside) {
            gr = a.dL_dcolor[pix];
            gg = a.dL_dcolor[HW + pix];
            gb = a.dL_dcolor[2 * HW + pix];
            if (a.dL_ddepth) gd = a.dL_ddepth[pix];
            if (a.dL_dalpha) ga = a.dL_dalpha[pix];
            T_final = a.iw.final_T[pix];
            last = (int)a.iw.n_contrib[pix];
        }
        // state at the END of this batch and at the forward's exit
        const float* ce = a.bw.ckpt + (size_t)(blockIdx.x + 1) * (5 * 64) + lane;
        const float* cf = a.bw.ckpt + (size_t)(b0 + (int)fe.y) * (5 * 64) + lane;
        float T = ce[0];
        const float inv_Te = __builtin_amdgcn_rcpf(T);
        // normalised suffix colour / depth behind this batch (zero once the pixel has finished)
        float rec_r = (cf[64] - ce[64]) * inv_Te, rec_g = (cf[128] - ce[128]) * inv_Te;
        float rec_b = (cf[192] - ce[192]) * inv_Te, rec_d = (cf[256] - ce[256]) * inv_Te;
        const float* __restrict__ bg = a.bg;
        // d/d(alpha_i) of [T_final * bg . g] and of [ga * (1 - T_final)]:  (T_final / (1 - alpha_i)) * (ga - bg.g)
        const float tail = T_final * (ga - (bg[0] * gr + bg[1] * gg + bg[2] * gb));
        float last_alpha = 0.f, lw_r = 0.f, lw_g = 0.f, lw_b = 0.f, lw_d = 0.f;
        wave_lds_fence();
        for (int k = min(cnt, n_eff - bstart) - 1; k >= 0; --k) {
            const int pos = bstart + k + 1;                   // 1-based list position
            const float4 g0 = s_g0[k];
            const float4 g1 = s_g1[k];
            const float dx = g0.x - fx, dy = g0.y - fy;
            const float p2 = gauss_power2(g1.x, g1.y, g1.z, dx, dy);
            const float G = gauss_falloff2(p2);
            const float alpha = fminf(ALPHA_MAX, g1.w * G);
            const bool contrib = inside && pos <= last && p2 <= 0.0f && alpha >= ALPHA_MIN;
            if (!__any(contrib)) continue;
            // Per-lane scalars; lanes that do not contribute get zeros through three selects (sG, aG, wgt) instead
            // of ten.  1 / (1 - alpha) is a hardware reciprocal (1 ulp) shared by the two divisions.
            float sG = 0.f, aG = 0.f, wgt = 0.f;
            float g2x = 0.f, g2y = 0.f, g2z = 0.f;
            if (contrib) {
                const float4 g2 = s_g2[k];
                g2x = g2.x; g2y = g2.y; g2z = g2.z;
                const float inv_one_m = __builtin_amdgcn_rcpf(1.0f - alpha);
                T = T * inv_one_m;
                wgt = alpha * T;
                rec_r = last_alpha * lw_r + (1.0f - last_alpha) * rec_r;
                rec_g = last_alpha * lw_g + (1.0f - last_alpha) * rec_g;
                rec_b = last_alpha * lw_b + (1.0f - last_alpha) * rec_b;
                rec_d = last_alpha * lw_d + (1.0f - last_alpha) * rec_d;
                lw_r = g2.x; lw_g = g2.y; lw_b = g2.z; lw_d = g0.z;
                last_alpha = alpha;
                float dL_dalpha = (g2.x - rec_r) * gr + (g2.y - rec_g) * gg + (g2.z - rec_b) * gb + (g0.z - rec_d) * gd;
                dL_dalpha = dL_dalpha * T + tail * inv_one_m;
                aG = G * dL_dalpha;                           // d/d(opacity)
                sG = g1.w * aG;                               // s = dL/dG * G
            }
            (void)g2x; (void)g2y; (void)g2z;
            float v[NACC];
            const float sdx = sG * dx, sdy = sG * dy;
            v[0] = sdx; v[1] = sdy;
            v[2] = sdx * dx; v[3] = sdx * dy; v[4] = sdy * dy;
            v[5] = aG;
            v[6] = wgt * gr; v[7] = wgt * gg; v[8] = wgt * gb; v[9] = wgt * gd;
            float q0, q1, q2;
            packed_reduce10(v, (lane & 1) != 0, (lane & 2) != 0, q0, q1, q2);
            // lanes 12..15 hold the totals: Partial layout {v0..v3 | v4..v7 | v8, v9, 0, 0}
            if ((lane & ~3) == 12) {
                float* o = reinterpret_cast<float*>(&s_out[k][0]) + (lane & 3);
                o[0] = q0;
                o[4] = q1;
                if ((lane & 2) == 0) o[8] = q2;
            }
        }
    }
    wave_lds_fence();
    if (lane < cnt) {
        float4* dst = reinterpret_cast<float4*>(a.partials) + (size_t)pslot * 3;
        dst[0] = s_out[lane][0];
        dst[1] = s_out[lane][1];
        dst[2] = s_out[lane][2];
    }
}

hipError_t launch_render_bwd(const RenderBwdArgs& a, hipStream_t s) {
    const uint64_t slots = a.capacity / BATCH;
    if (slots == 0) return hipSuccess;
    render_bwd_kernel<<<(unsigned)slots, RBLOCK, 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace exa
