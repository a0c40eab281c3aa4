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
// Why it looks the way it does (measured on MI355X, tools/probe + EXA_PROBE_FWD builds):
//  * the render kernels are bound by VALU ISSUE SLOTS, not memory: a wave64 VALU op occupies its SIMD for 4
//    cycles, so (waves x splats x instructions) / 1024 SIMDs is the floor.  Hence
//  * splats are evaluated two at a time on packed fp32 (v_pk_add/mul/fma_f32: same 4 cycles, two results),
//    which needs the staged batch in SoA form (px[64], py[64], ...) so that a pair is one aligned 8 bytes;
//  * per-pixel state is arithmetic (`live` = 1.0f / 0.0f), never boolean: boolean state becomes SGPR-mask
//    traffic (v_cmp -> s_and/s_or -> v_cndmask) that doubled the instruction count of the blend.
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

// Stage the record held by this lane (rows 0..2 of its Splat) as entry `lane` of the batch.
__device__ __forceinline__ void stage_splat(BatchLds& s, int lane, const float4& r0, const float4& r1, const float4& r2) {
    s.px[lane] = r0.x; s.py[lane] = r0.y;
    s.ca[lane] = r1.x; s.cb[lane] = r1.y; s.cc[lane] = r1.z; s.op[lane] = r1.w;
    s.col[lane] = make_float4(r2.x, r2.y, r2.z, r0.z);
}

// alpha (0 when the splat is skipped at this pixel: power > 0 or alpha < 1/255) and falloff G of two splats
struct Alpha2 { v2f alpha, G; };
__device__ __forceinline__ Alpha2 splat_alpha2(v2f px, v2f py, v2f ca, v2f cb, v2f cc, v2f op, float fx, float fy) {
#pragma clang fp contract(off)
    const v2f dx = px - fx, dy = py - fy;
    const v2f q = __builtin_elementwise_fma(cc * dy, dy, (ca * dx) * dx);
    const v2f p2 = __builtin_elementwise_fma(-(cb * dx), dy, -0.5f * q);     // log2 of the falloff
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

// alphas (and falloffs) of splats k .. k+3 of the staged batch at pixel (fx, fy); k is a multiple of 4
struct Alpha4 { float alpha[4], G[4]; };
__device__ __forceinline__ Alpha4 splat_alpha4(const BatchLds& s, int k, float fx, float fy) {
    const v4f px = *reinterpret_cast<const v4f*>(&s.px[k]);
    const v4f py = *reinterpret_cast<const v4f*>(&s.py[k]);
    const v4f ca = *reinterpret_cast<const v4f*>(&s.ca[k]);
    const v4f cb = *reinterpret_cast<const v4f*>(&s.cb[k]);
    const v4f cc = *reinterpret_cast<const v4f*>(&s.cc[k]);
    const v4f op = *reinterpret_cast<const v4f*>(&s.op[k]);
    const Alpha2 lo = splat_alpha2(px.xy, py.xy, ca.xy, cb.xy, cc.xy, op.xy, fx, fy);
    const Alpha2 hi = splat_alpha2(px.zw, py.zw, ca.zw, cb.zw, cc.zw, op.zw, fx, fy);
    Alpha4 r;
    r.alpha[0] = lo.alpha.x; r.alpha[1] = lo.alpha.y; r.alpha[2] = hi.alpha.x; r.alpha[3] = hi.alpha.y;
    r.G[0] = lo.G.x; r.G[1] = lo.G.y; r.G[2] = hi.G.x; r.G[3] = hi.G.y;
    return r;
}
// tail of a batch (k not a multiple of 4 away from the end): one splat
__device__ __forceinline__ void splat_alpha1(const BatchLds& s, int k, float fx, float fy, float& alpha, float& G) {
    const float one = 1.0f;
    const Alpha2 r = splat_alpha2(v2f{s.px[k], one}, v2f{s.py[k], one}, v2f{s.ca[k], one}, v2f{s.cb[k], one},
                                  v2f{s.cc[k], one}, v2f{s.op[k], 0.0f}, fx, fy);
    alpha = r.alpha.x;
    G = r.G.x;
}

// One step of the front-to-back recurrence.  `live` is 1.0f while the pixel accepts splats, 0.0f after it
// stopped (or for a pixel outside the image).  Returns the blend weight w = a T (0 when nothing is blended)
// and the effective alpha `a`; updates T and live.  Exactly the sequential rule: a dead pixel and a skipped
// splat (alpha == 0) blend nothing; the splat that would push T (1 - a) under 1e-4 is NOT blended and kills
// the pixel.  (T >= 1e-4 is an invariant, so a == 0 can never trigger the stop.)
__device__ __forceinline__ float blend_step(float& T, float& live, float alpha, float& a) {
#pragma clang fp contract(off)
    a = alpha * live;                                 // exact
    const float tT = __builtin_fmaf(-a, T, T);        // T (1 - a)
    const bool stop = tT < T_EPS;
    const float w = stop ? 0.0f : a * T;
    live = stop ? 0.0f : live;
    T = stop ? T : tT;
    return w;
}

}  // namespace exa
