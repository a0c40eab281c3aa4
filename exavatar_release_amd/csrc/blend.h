// The per-(pixel, splat) arithmetic shared by render_fwd.hip and render_bwd.hip.
//
// The backward pass REPLAYS the forward recurrence of a 64-splat batch from the checkpointed state at the
// batch start, so both kernels must take bit-identical decisions (skip alpha < 1/255, stop when
// T (1 - alpha) < 1e-4).  Everything that feeds those decisions lives here, is written with explicit fma /
// mul / add (contraction off) and is compiled into both kernels from this one source.
//
// Follows oracle/raster_oracle.py steps 9/10 (the per-pixel rule of the rasterizer behind reference
// avatar/common/nets/module.py:632-640).
//
// Why it looks the way it does (measured on MI355X: tools/probe/valu_probe.hip, EXA_PROBE_FWD builds, SQ PMC
// passes in profiles/):
//  * the render kernels are bound by per-wave LATENCY and VALU issue, not by memory: only ~3.7 waves per SIMD
//    exist (one per non-empty 8x8 sub-tile), VALU-active is 55-60 % of the wave cycles.  One SIMD retires a
//    plain fp32 VALU op per ~2.5 cycles, a packed v_pk_*_f32 per ~5 (two results: same throughput, half the
//    issue slots), v_exp_f32 per ~8, and a broadcast ds_read_b128 costs 4 LDS cycles of the whole CU.  Hence
//  * the staged batch is SoA (px[64], py[64], ...): four splats are ONE ds_read_b128 per field, and pairs of
//    splats form packed operands;
//  * per-pixel state is arithmetic (`live` = 1.0f / 0.0f), never boolean: boolean state becomes SGPR-mask
//    traffic (v_cmp -> s_and/s_or -> v_cndmask) that doubled the instruction count of the blend;
//  * the recurrence is advanced four splats at a time through partial products (blend_group4), so the
//    loop-carried dependency is one multiply per group.
#pragma once
#include "common.h"

namespace exa {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// One batch of 64 splats staged in LDS, SoA so that splats (k, k + 1) form packed operands and four splats
// are one ds_read_b128 (all lanes read the same address: broadcast).
struct BatchLds {
    float px[BATCH], py[BATCH], ca[BATCH], cb[BATCH], cc[BATCH], op[BATCH];
    float4 col[BATCH];                           // r, g, b, depth
};

// Stage the record held by this lane (rows 0..2 of its Splat) as entry `lane` of the batch.  Row 2 = (r, g, b, depth) goes
// to LDS as loaded (common.h, Splat).
// (r0 = the first TWO words of row 0 only: with the whole row loaded, the compiler recycled the registers of its dead
//  components as address temporaries right behind the prefetch and waited for the load to land before it could.)
__device__ __forceinline__ void stage_splat(BatchLds& s, int lane, const float2& r0, const float4& r1, const float4& r2) {
    s.px[lane] = r0.x; s.py[lane] = r0.y;
    s.ca[lane] = r1.x; s.cb[lane] = r1.y; s.cc[lane] = r1.z; s.op[lane] = r1.w;
    s.col[lane] = r2;
}

// alpha (0 when the splat is skipped at this pixel: power > 0 or alpha < 1/255) and falloff G of two splats
struct Alpha2 { v2f alpha, G; };
__device__ __forceinline__ Alpha2 splat_alpha2(v2f px, v2f py, v2f ca, v2f cb, v2f cc, v2f op, float fx, float fy) {
#pragma clang fp contract(off)
    const v2f dx = px - fx, dy = py - fy;
    // (ca, cb, cc) = (-A/2, -B, -C/2) log2(e): log2 of the falloff in four multiplies and two fmas
    const v2f p2 = __builtin_elementwise_fma(ca * dx, dx, __builtin_elementwise_fma(cc * dy, dy, (cb * dx) * dy));
    Alpha2 r;
    r.G.x = __builtin_amdgcn_exp2f(p2.x);
    r.G.y = __builtin_amdgcn_exp2f(p2.y);
    const v2f og = op * r.G;
    float a0 = fminf(ALPHA_MAX, og.x), a1 = fminf(ALPHA_MAX, og.y);
    a0 = (a0 >= ALPHA_MIN) ? a0 : 0.0f;
    a1 = (a1 >= ALPHA_MIN) ? a1 : 0.0f;
    r.alpha.x = (p2.x <= 0.0f) ? a0 : 0.0f;
    r.alpha.y = (p2.y <= 0.0f) ? a1 : 0.0f;
    return r;
}

// Operands of four consecutive splats k .. k+3 (k a multiple of 4): six broadcast ds_read_b128.
struct Ops4 { v4f px, py, ca, cb, cc, op; };
__device__ __forceinline__ Ops4 load_ops4(const BatchLds& s, int k) {
    Ops4 o;
    o.px = *reinterpret_cast<const v4f*>(&s.px[k]);
    o.py = *reinterpret_cast<const v4f*>(&s.py[k]);
    o.ca = *reinterpret_cast<const v4f*>(&s.ca[k]);
    o.cb = *reinterpret_cast<const v4f*>(&s.cb[k]);
    o.cc = *reinterpret_cast<const v4f*>(&s.cc[k]);
    o.op = *reinterpret_cast<const v4f*>(&s.op[k]);
    return o;
}
// alphas (and falloffs) of four splats at pixel (fx, fy)
struct Alpha4 { float alpha[4], G[4]; };
__device__ __forceinline__ Alpha4 splat_alpha4(const Ops4& o, float fx, float fy) {
    const Alpha2 lo = splat_alpha2(o.px.xy, o.py.xy, o.ca.xy, o.cb.xy, o.cc.xy, o.op.xy, fx, fy);
    const Alpha2 hi = splat_alpha2(o.px.zw, o.py.zw, o.ca.zw, o.cb.zw, o.cc.zw, o.op.zw, fx, fy);
    Alpha4 r;
    r.alpha[0] = lo.alpha.x; r.alpha[1] = lo.alpha.y; r.alpha[2] = hi.alpha.x; r.alpha[3] = hi.alpha.y;
    r.G[0] = lo.G.x; r.G[1] = lo.G.y; r.G[2] = hi.G.x; r.G[3] = hi.G.y;
    return r;
}

// Four steps of the front-to-back recurrence at once.  `live` is 1.0f while the pixel accepts splats, 0.0f
// after it stopped (or for a pixel outside the image).  Outputs per splat j: the effective alpha a[j], the
// transmittance in front of it Tb[j] and the blend weight w[j] = a[j] Tb[j] (0 when nothing is blended).
//
// The sequential rule -- skip alpha == 0, and the splat that would push T (1 - a) under 1e-4 is NOT blended
// and kills the pixel -- is evaluated through the partial products P_j = prod_{i <= j} (1 - a_i), which depend
// on the alphas only: the loop-carried dependency is ONE multiply (T P_4) per four splats instead of a
// mul / fma / compare / select chain per splat (the kernels are latency bound: ~3.7 waves per SIMD).  T P_j is
// non-increasing in j, so "stopped at or before j" is simply T P_j < 1e-4; a dead pixel has a = 0, P = 1.
// T >= 1e-4 is an invariant.
__device__ __forceinline__ void blend_group4(float& T, float& live, const float (&alpha)[4], float (&a)[4],
                                             float (&Tb)[4], float (&w)[4]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = alpha[j] * live;          // exact
    const float P1 = 1.0f - a[0];
    const float P2 = P1 * (1.0f - a[1]);
    const float P3 = P2 * (1.0f - a[2]);
    const float P4 = P3 * (1.0f - a[3]);
    Tb[0] = T; Tb[1] = T * P1; Tb[2] = T * P2; Tb[3] = T * P3;
    const float T4 = T * P4;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = a[j] * Tb[j];
    if (__any(T4 < T_EPS)) {                                     // some pixel stops inside this group (rare)
        const float after[4] = {Tb[1], Tb[2], Tb[3], T4};
        float Tn = T;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool stop = after[j] < T_EPS;
            w[j] = stop ? 0.0f : w[j];
            Tn = stop ? Tn : after[j];
        }
        live = (T4 < T_EPS) ? 0.0f : live;
        T = Tn;
    } else {
        T = T4;
    }
}

}  // namespace exa
