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
//  * the render kernels are bound by VALU issue (and, below ~4 waves per SIMD, by per-wave latency), not by memory: only
//    ~3.7 waves per SIMD exist in the forward (one per non-empty 8x8 sub-tile).  One SIMD retires a
//    plain fp32 VALU op per ~2.5 cycles, a packed v_pk_*_f32 per ~5 (two results: same throughput, half the
//    issue slots), v_exp_f32 per ~8, and a broadcast ds_read_b128 costs 4 LDS cycles of the whole CU.  Hence
//  * the staged batch is SoA (px[64], py[64], ...): four splats are ONE ds_read_b128 per field, and pairs of
//    splats form packed operands;
//  * per-pixel state is arithmetic (a stopped pixel has T = 0.0f), never boolean: boolean state becomes SGPR-mask
//    traffic (v_cmp -> s_and/s_or -> v_cndmask) that doubled the instruction count of the blend.
#pragma once
#include "common.h"

namespace exa {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// One batch of 64 splats staged in LDS, SoA so that splats (k, k + 1) form packed operands and four splats
// are one ds_read_b128 (all lanes read the same address: broadcast).
struct BatchLds {
    float px[BATCH], py[BATCH], ca[BATCH], cb[BATCH], cc[BATCH], op[BATCH];      // op: log2(opacity), see splat_alpha2
    float4 col[BATCH];                           // r, g, b, depth
};

// Stage the record held by this lane (rows 0..2 of its Splat) as entry `lane` of the batch.  Row 2 = (r, g, b, depth) goes
// to LDS as loaded (common.h, Splat).
// (r0 = the first TWO words of row 0 only: with the whole row loaded, the compiler recycled the registers of its dead
//  components as address temporaries right behind the prefetch and waited for the load to land before it could.)
__device__ __forceinline__ void stage_splat(BatchLds& s, int lane, const float2& r0, const float4& r1, const float4& r2) {
    s.px[lane] = r0.x; s.py[lane] = r0.y;
    s.ca[lane] = r1.x; s.cb[lane] = r1.y; s.cc[lane] = r1.z; s.op[lane] = __builtin_amdgcn_logf(r1.w);     // v_log_f32: log2; 0 -> -inf
    s.col[lane] = r2;
}

// "power > 0 -> skip" (upstream's guard against conics that are not positive definite) can only fire for a conic that is
// indefinite, or so ill-conditioned that fp32 loses the sign of the quadratic form: the Horner sum below errs by <= ~2e-7
// |trace| r^2, a definite form is >= lambda_min r^2 in magnitude, and lambda_min / lambda_max >= det / trace^2.  (ca, cb, cc) as
// staged (scaled, NEGATIVE definite: [[ca, cb/2], [cb/2, cc]]); the all-zero record staged behind the end of a list counts as
// safe (its sum is lop = -inf exactly).  Both blends vote once per staged batch; a group of four WITH an unsafe entry (needles
// beyond ~600 : 1, broken covariances -- never an avatar or a scene splat) takes the guard as a fix-up behind the evaluation
// (power_guard4), every other group evaluates without it: two compares and a scalar AND less per splat and pixel.
__device__ __forceinline__ bool conic_safe(float ca, float cb, float cc) {
    const float tr = ca + cc;
    return ca <= 0.0f && cc <= 0.0f && 4.0f * ca * cc - cb * cb >= 1e-5f * (tr * tr);
}

// alpha (0 when the splat is skipped at this pixel: alpha < 1/255; power > 0: power_guard4) and A = opacity x falloff (alpha
// before the clamp to 0.99: what the backward pass differentiates, straight through the clamp) of two splats.
// (ca, cb, cc) = (-A/2, -B, -C/2) log2(e) and lop = log2(opacity):  p2 = log2(opacity G) = dx (ca dx + cb dy) + (cc dy) dy + lop
// in three multiplies and two fmas per splat (rounds 2-5: four and two, and a multiply by the opacity behind the exponential),
// and "power <= 0" is "p2 <= lop".
// vis: the compare behind "NOT skipped at this pixel" -- an SGPR mask whose ballot is the compare's own result.
// Ag = A where a LIVE pixel takes the splat, else 0 (`live`: the caller's compare T > 0 at the start of the group; selected
// here, in the basic block of the compares -- a compare result that crosses a branch comes back through a VGPR).
struct Alpha2 { v2f alpha, A, Ag, p2; bool vis0, vis1; };
__device__ __forceinline__ Alpha2 splat_alpha2(v2f px, v2f py, v2f ca, v2f cb, v2f cc, v2f lop, float fx, float fy, bool live) {
#pragma clang fp contract(off)
    const v2f dx = px - fx, dy = py - fy;
    const v2f t = __builtin_elementwise_fma(cb, dy, ca * dx);
    const v2f u = __builtin_elementwise_fma(cc * dy, dy, lop);
    Alpha2 r;
    r.p2 = __builtin_elementwise_fma(t, dx, u);
    r.A.x = __builtin_amdgcn_exp2f(r.p2.x);
    r.A.y = __builtin_amdgcn_exp2f(r.p2.y);
    const float a0 = fminf(ALPHA_MAX, r.A.x), a1 = fminf(ALPHA_MAX, r.A.y);
    r.vis0 = a0 >= ALPHA_MIN;
    r.vis1 = a1 >= ALPHA_MIN;
    r.alpha.x = r.vis0 ? a0 : 0.0f;
    r.alpha.y = r.vis1 ? a1 : 0.0f;
    r.Ag.x = (r.vis0 && live) ? r.A.x : 0.0f;
    r.Ag.y = (r.vis1 && live) ? r.A.y : 0.0f;
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
// alphas (and unclamped opacity x falloff) of four splats at pixel (fx, fy); took = wave masks of "this pixel takes the splat"
struct Alpha4 { float alpha[4], Ag[4], p2[4]; unsigned long long took[4]; };
__device__ __forceinline__ Alpha4 splat_alpha4(const Ops4& o, float fx, float fy, bool live) {
    const Alpha2 lo = splat_alpha2(o.px.xy, o.py.xy, o.ca.xy, o.cb.xy, o.cc.xy, o.op.xy, fx, fy, live);
    const Alpha2 hi = splat_alpha2(o.px.zw, o.py.zw, o.ca.zw, o.cb.zw, o.cc.zw, o.op.zw, fx, fy, live);
    Alpha4 r;
    r.alpha[0] = lo.alpha.x; r.alpha[1] = lo.alpha.y; r.alpha[2] = hi.alpha.x; r.alpha[3] = hi.alpha.y;
    r.Ag[0] = lo.Ag.x; r.Ag[1] = lo.Ag.y; r.Ag[2] = hi.Ag.x; r.Ag[3] = hi.Ag.y;
    r.p2[0] = lo.p2.x; r.p2[1] = lo.p2.y; r.p2[2] = hi.p2.x; r.p2[3] = hi.p2.y;
    // (the intrinsic, not __ballot: HIP's wrapper compares an int against 0, i.e. takes the mask through a VGPR and back)
    r.took[0] = __builtin_amdgcn_ballot_w64(lo.vis0); r.took[1] = __builtin_amdgcn_ballot_w64(lo.vis1);
    r.took[2] = __builtin_amdgcn_ballot_w64(hi.vis0); r.took[3] = __builtin_amdgcn_ballot_w64(hi.vis1);
    return r;
}
// The guard of a group with an unsafe conic (conic_safe): power > 0 -> the splat is skipped at this pixel.
__device__ __forceinline__ void power_guard4(Alpha4& e, const Ops4& o) {
    const float lop[4] = {o.op.x, o.op.y, o.op.z, o.op.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool neg = e.p2[j] <= lop[j];
        e.alpha[j] = neg ? e.alpha[j] : 0.0f;
        e.Ag[j] = neg ? e.Ag[j] : 0.0f;
        e.took[j] &= __builtin_amdgcn_ballot_w64(neg);
    }
}

// Four steps of the front-to-back recurrence.  State per pixel: T = its transmittance while it accepts splats, 0.0f once it
// has stopped (or for a pixel outside the image) -- a stopped pixel then takes w = alpha * 0 = 0 of everything that follows
// without a `live` factor -- and Tdead = the transmittance it stopped with (what the image and the checkpoints report).
// Outputs per splat j: the transmittance in front of it Tb[j] and the blend weight w[j] = alpha[j] Tb[j] (0 when nothing is
// blended).  A splat CHANGES a pixel's state -- is blended, or stops it: what the backward pass has to replay -- exactly when its
// alpha is not skipped (Alpha4::took) and the pixel was live at the start of the group: inside a group T_j > 0 <=> T > 0
// (alpha <= 0.99), and entries behind the one that stops the pixel are flagged too, which is harmless.
//
// The sequential rule -- skip alpha == 0, and the splat that would push T (1 - alpha) under 1e-4 is NOT blended and kills the
// pixel -- as  w_j = alpha_j T_j,  T_{j+1} = T_j - w_j:  two plain instructions per splat.  (Rounds 2-5 advanced four splats
// through the partial products P_j = prod (1 - a_i) so that the loop-carried dependency was ONE multiply per group -- 19
// instructions per group instead of 8, worth it at the 3.7 waves per SIMD of those rounds; at five the dependent chain of
// eight hides behind the other waves: with the Horner power and the scalar votes, render_fwd 37.2 -> 31.2 us on C3.)  A skipped splat leaves T bit-for-bit
// unchanged, so the backward pass, which replays the COMPACTED list (other groups of four), reproduces the forward's
// transmittances exactly.  T_j is non-increasing in j, so "stopped at or before j" is simply T_{j+1} < 1e-4; alpha <= 0.99
// keeps a live T above 1e-6 T_j > 0, and T >= 1e-4 is an invariant of a live pixel.
// gate[j]: a per-splat factor of the caller's that has to vanish with w[j] when the stop rule zeroes it (render_bwd.hip).
// live_mask: the caller's ballot of T > 0 at the start of the group (taken in the block of that compare: as a bool it
// would cross the guard's branch and come back through a VGPR).
__device__ __forceinline__ void blend_group4(float& T, float& Tdead, unsigned long long live_mask, const float (&alpha)[4],
                                             float (&Tb)[4], float (&w)[4], float (&gate)[4]) {
#pragma clang fp contract(off)
    Tb[0] = T;
    w[0] = alpha[0] * Tb[0]; Tb[1] = Tb[0] - w[0];
    w[1] = alpha[1] * Tb[1]; Tb[2] = Tb[1] - w[1];
    w[2] = alpha[2] * Tb[2]; Tb[3] = Tb[2] - w[2];
    w[3] = alpha[3] * Tb[3];
    const float T4 = Tb[3] - w[3];
    if ((__builtin_amdgcn_ballot_w64(T4 < T_EPS) & live_mask) != 0ull) {      // some live pixel stops inside this group (rare)
        const float after[4] = {Tb[1], Tb[2], Tb[3], T4};
        const bool dies = T4 < T_EPS && T > 0.0f;
        float Tn = T;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                            // (a pixel that was dead already: all of it 0)
            const bool stop = after[j] < T_EPS;
            w[j] = stop ? 0.0f : w[j];
            gate[j] = stop ? 0.0f : gate[j];
            Tn = stop ? Tn : after[j];
        }
        Tdead = dies ? Tn : Tdead;
        T = (T4 < T_EPS) ? 0.0f : T4;
    } else {
        T = T4;
    }
}
__device__ __forceinline__ void blend_group4(float& T, float& Tdead, unsigned long long live_mask, const float (&alpha)[4],
                                             float (&Tb)[4], float (&w)[4]) {
    float none[4] = {0.f, 0.f, 0.f, 0.f};
    blend_group4(T, Tdead, live_mask, alpha, Tb, w, none);
}

}  // namespace exa
