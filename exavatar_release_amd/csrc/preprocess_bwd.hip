// Per-Gaussian backward for gfx950: gathers the per-instance `Partial` sums render_bwd.hip wrote
// (Gaussian-major, contiguous per Gaussian -> no atomics) and applies the chain rule down to
// dL/d{means3D, means2D, scales, rotations, opacities, colours | SH, cov3D}.
//
// Replaces upstream computeCov2DCUDA (backward) + preprocessCUDA (backward) of the rasterizer behind
// reference avatar/common/nets/module.py:632-640.  The chain rule is derived from the forward in
// oracle/raster_oracle.py::preprocess (same symbols: pv*, h*, S**, T**, J**, a/b/c, conic) and checked
// against its autograd.  Semantics kept from upstream: the +-1.3 tanfov clamp blocks the gradient
// through t.x / t.y when active; quaternions are not normalised; dL/dmeans2D is reported in NDC units
// (pixel gradient * (W/2, H/2), z = 0), which is what the reference's densification reads
// (avatar/main/train.py:51).
//
// HBM traffic per Gaussian: reads 1 B per instance + 40 B per BLENDED instance + 32 B of the splat record + 44 B
// inputs, writes 68 B of gradients (SH: + 12 * M B).
#include "common.h"

namespace exa {

// One thread per (job, Gaussian).  SUM = false: gridDim.y jobs, thread = (job blockIdx.y, Gaussian idx).
// SUM = true: the K jobs are K views of the SAME Gaussians and job 0's outputs receive the summed gradient
// (dL_dmeans2D stays per view: the densification statistics need per-view norms, reference
// avatar/common/nets/module.py:155-157).  VW = 4 ("views over waves"): a workgroup of vw = min(K, 4) waves shares 64
// Gaussians, every wave takes ONE view (no loop: wave w of workgroup row g = blockIdx.y takes view g * vw + w) and the
// waves add their results through LDS (the staging area of the gather, dead by then).  K > 4: the second row of
// workgroups writes ITS sum into a scratch behind the partial records of its first view (PreprocessBwdArgs.group_scratch)
// and `sum_groups_kernel` adds it to the outputs.  Until round 6 the four waves of a 256-thread workgroup LOOPED over the
// views (w, w + 4): K = 2 left half of every workgroup idle while it held 74 KB of LDS, 64-66 us against 2 x 19 us for
// two single-view launches; now K views cost what K single launches cost, in one.  VW = 1 (the SH path, whose 48-float
// dL_dsh rows are accumulated in global memory by their single owner): one thread walks all views.  SH = false compiles the SH backward out (the
// colours-precomp path of ExAvatar's renderer, module.py:632-640): fewer registers, more waves per SIMD.
// PREFIX = true: the job may have a constant prefix (grad_first > 0); its own instantiation so that the plain kernels
// carry none of it.
// dens_shared != 0 (SUM mode only): all K views update the SAME densification-statistics arrays (K views of one model
// accumulated into one set of statistics): the per-view norms / counts / radii are summed in registers, across the
// waves through LDS, and written by ONE thread per Gaussian -- the per-view read-modify-write below would race between
// the waves of a workgroup.  (Distinct arrays per view need no such care; partial aliasing is rejected by the C ABI.)
// ---- wave-cooperative gather of the partial records ---------------------------------------------------------------
// The 64 Gaussians of a wave own CONSECUTIVE ranges of the Gaussian-major instance numbering (a typical avatar splat ~6
// instances, some lane of most waves 25-40).  Until round 4 every lane walked its own range -- `touched` bytes, then the
// flagged records, eight instances per trip -- and the wave left that loop with its slowest lane: five trips of two
// dependent, fully divergent gathers = 9-12 us of the kernel's 17.6 on C3 (tools/gpu_pbwd_phases.py).  Now the wave
// streams the concatenation of its lanes' ranges through LDS: stream position q -> (lane, instance) by a binary search
// over the wave's prefix of instance counts, GCH positions per chunk with ALL their `touched` bytes requested at once and
// then all their flagged records (neighbouring lanes read neighbouring slots: coalesced), staged in LDS; each lane then
// sums ITS instances from LDS in slot order -- the very order and arithmetic of the old per-lane loop (an unflagged slot adds
// +0), so the results are bit-identical to it.  Two dependent global round trips per GCH slots of the whole wave instead of
// per eight instances of its slowest lane.  Splats with >= COOP_MIN instances keep their own whole-wave path.
#ifndef EXA_PBWD_GCH
#define EXA_PBWD_GCH 256
#endif
constexpr int GCH = EXA_PBWD_GCH;             // staged slots per chunk and wave (48 B of LDS each)
struct GatherLds {
    uint32_t pre[64], off[64];
    float4 rec[GCH * 3];
};
// The flagged records are fetched with BUFFER loads whose offset is pushed out of range for an unflagged slot: the hardware
// returns zeros for such a lane without a memory request -- predication without a branch, so that the twelve loads of a
// chunk leave back to back (conditional loads cost one `s_waitcnt` per branch region).
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr uint32_t BUF_OOB = 0xffffff00u;          // + 32 does not wrap; >= any bound this path is used with
__device__ __forceinline__ void stream_gather(uint32_t off, uint32_t n, const uint8_t* __restrict__ touched,
                                              const float4* __restrict__ prec, uint64_t prec_bytes, float (&sum)[10],
                                              GatherLds& L, int lane) {
    const bool use_buf = prec_bytes < (uint64_t)BUF_OOB;               // (a 4 GiB record array takes the pointer path: 32-bit byte offsets)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float4*>(prec), 0, use_buf ? (int)(uint32_t)prec_bytes : 0, 0x00020000);
    uint32_t incl = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += o;
    }
    const uint32_t pre = incl - n;
    const uint32_t S = (uint32_t)__shfl((int)incl, 63, 64);            // wave-uniform
    if (S == 0u) return;
    L.pre[lane] = pre; L.off[lane] = off;
    wave_lds_fence();
    for (uint32_t c0 = 0; c0 < S; c0 += GCH) {
        // (Every loop below is its own straight-line block ON PURPOSE: with a load and its first use -- or a conditional
        //  load and the store of its result -- in one loop body the compiler emits one branch region per iteration and an
        //  `s_waitcnt vmcnt(0)` at each join: eight serial round trips per chunk instead of two.)
        uint32_t slot[GCH / 64], tch[GCH / 64];
        bool in[GCH / 64];
#pragma unroll
        for (int u = 0; u < GCH / 64; ++u) {
            const uint32_t q = c0 + 64u * u + (uint32_t)lane;
            in[u] = q < S;
            // the LAST lane g with pre[g] <= q owns position q (lanes without instances share their successor's prefix)
            uint32_t g = 0;
#pragma unroll
            for (int s2 = 32; s2 > 0; s2 >>= 1)
                if (L.pre[g + s2] <= q) g += s2;
            slot[u] = in[u] ? L.off[g] + (q - L.pre[g]) : off;          // (past the end: any address that is valid to read)
        }
#pragma unroll
        for (int u = 0; u < GCH / 64; ++u) tch[u] = touched[slot[u]];   // unconditional: all requests leave back to back
        uint32_t fl = 0;
#pragma unroll
        for (int u = 0; u < GCH / 64; ++u) fl |= (in[u] && tch[u]) ? (1u << u) : 0u;
        float4 q0[GCH / 64], q1[GCH / 64], q2[GCH / 64];
#pragma unroll
        for (int u = 0; u < GCH / 64; ++u) {
            q0[u] = make_float4(0.f, 0.f, 0.f, 0.f); q1[u] = q0[u]; q2[u] = q0[u];
        }
        if (use_buf) {
#pragma unroll
            for (int u = 0; u < GCH / 64; ++u) {                        // only the flagged records are fetched
                const uint32_t bo = ((fl >> u) & 1u) ? slot[u] * (uint32_t)PARTIAL_BYTES : BUF_OOB;
                q0[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, bo, 0, 0));
                q1[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, bo + 16u, 0, 0));
                if (PARTIAL_BYTES == 40) {
                    const float2 t = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, bo + 32u, 0, 0));
                    q2[u] = make_float4(t.x, t.y, 0.f, 0.f);
                } else {
                    q2[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, bo + 32u, 0, 0));
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < GCH / 64; ++u) {
                if ((fl >> u) & 1u) partial_load(prec, slot[u], q0[u], q1[u], q2[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < GCH / 64; ++u) {
            float4* dst = L.rec + (64 * u + lane) * 3;
            dst[0] = q0[u]; dst[1] = q1[u]; dst[2] = q2[u];
        }
        wave_lds_fence();
        const uint32_t lo = pre > c0 ? pre : c0, hi = (pre + n) < (c0 + GCH) ? (pre + n) : (c0 + GCH);
        for (uint32_t j = lo; j < hi; ++j) {
            const float4* src = L.rec + (j - c0) * 3;
            const float4 q0 = src[0], q1 = src[1], q2 = src[2];
            sum[0] += q0.x; sum[1] += q0.y; sum[2] += q0.z; sum[3] += q0.w;
            sum[4] += q1.x; sum[5] += q1.y; sum[6] += q1.z; sum[7] += q1.w;
            sum[8] += q2.x; sum[9] += q2.y;
        }
        wave_lds_fence();                                           // the next chunk overwrites the staging area
    }
}

// ||dL/dmean2D|| of the densification statistics: ONE spelling (an explicit fma) for the kernel that accumulates it inside the
// backward pass and the stand-alone one, which have to agree bit for bit (tests/test_gpu_parity.py, fused densify statistics) --
// left to the compiler, x * x + y * y is contracted in one kernel and not in the other depending on the code around it.
__device__ __forceinline__ float grad_norm2(float x, float y) { return sqrtf(fmaf(x, x, y * y)); }

// cov2D = T Sigma T^T, its determinant and the gradient of the conic: in DOUBLE since round 5 (everything else float).
// A needle-like projected covariance (eigenvalues 894 and 1.0 px^2 in the seeds that showed it: x300 scales, unnormalised
// quaternions) makes (da, db, dc) differences of terms 200^2 x their size, so the float rounding of a, b, c
// themselves (1e-7 relative, accumulated over the two 3 x 3 products) came back as up to 3 % of that Gaussian's
// dL/dscale: 1.5e-3 of the tensor's max-norm, 2 of 300 fuzz seeds over the 1e-3 bar (EXA_FUZZ_TRIALS=300).  Which
// stage has to be double was settled on the C oracle with one precision per stage (mean projection, Sigma, J / T,
// T Sigma T^T, conic gradient, everything behind): T Sigma T^T + the conic gradient alone bring the worst of the 300 seeds
// to 3.4e-5, the same as all-double; any other single stage leaves it at 1.2-2.2e-3.  (All-double was built first:
// 134 -> 168 VGPRs here, 248 -> 283 in the SUM kernels = one wave per SIMD there, K = 8 batches 9 250 -> 8 370 it/s.)
struct Cov2DGrad { float ST0[3], ST1[3], dpx, dpy, da, db, dc; };
__device__ __forceinline__ Cov2DGrad cov2d_conic_grad(float S00, float S01, float S02, float S11, float S12, float S22,
                                                      const float (&T0)[3], const float (&T1)[3],
                                                      float mx, float my, float mxx, float mxy, float myy) {
#pragma clang fp contract(off)
    const double t0x = T0[0], t0y = T0[1], t0z = T0[2], t1x = T1[0], t1y = T1[1], t1z = T1[2];
    const double ST0[3] = {S00 * t0x + S01 * t0y + S02 * t0z, S01 * t0x + S11 * t0y + S12 * t0z, S02 * t0x + S12 * t0y + S22 * t0z};
    const double ST1[3] = {S00 * t1x + S01 * t1y + S02 * t1z, S01 * t1x + S11 * t1y + S12 * t1z, S02 * t1x + S12 * t1y + S22 * t1z};
    const double a = (t0x * ST0[0] + t0y * ST0[1] + t0z * ST0[2]) + LOWPASS;
    const double b = t0x * ST1[0] + t0y * ST1[1] + t0z * ST1[2];
    const double c = (t1x * ST1[0] + t1y * ST1[1] + t1z * ST1[2]) + LOWPASS;
    const double det = a * c - b * b;
    // upstream computeCov2DCUDA (backward): "denom2inv" = 1 / (det^2 + 1e-7), not the exact 1 / det^2
    const double idet = 1.0 / det, idet2 = 1.0 / (det * det + 1e-7);
    // moments of s = dL/dG * G  ->  d/d(pixel centre) and d/d(conic) with the raw conic (A, B, C) = (c, -b, a) / det
    const double cA = c * idet, cB = -b * idet, cC = a * idet;
    const double dA = -0.5 * mxx, dB = -(double)mxy, dC = -0.5 * myy;
    Cov2DGrad r;
#pragma unroll
    for (int j = 0; j < 3; ++j) { r.ST0[j] = (float)ST0[j]; r.ST1[j] = (float)ST1[j]; }
    r.dpx = (float)(-cA * mx - cB * my);
    r.dpy = (float)(-cC * my - cB * mx);
    // conic -> cov2D (a, b, c)
    r.da = (float)(idet2 * (-c * c * dA + b * c * dB - b * b * dC));
    r.dc = (float)(idet2 * (-b * b * dA + b * a * dB - a * a * dC));
    r.db = (float)(idet2 * (2.0 * b * c * dA - (det + 2.0 * b * b) * dB + 2.0 * a * b * dC));
    return r;
}

template <int N> struct StaticGather {
    __device__ static __forceinline__ GatherLds* get() { __shared__ GatherLds g[N]; return g; }
};
template <> struct StaticGather<0> {
    __device__ static __forceinline__ GatherLds* get() { return nullptr; }
};

// (two waves per SIMD at least: the SUM kernels sit at the 256-register line, and one wave per SIMD costs the K = 8 batches 5-10 %)
template <bool SUM, int VW, bool SH, bool PREFIX>
__global__ __launch_bounds__(BLOCK, 2) void preprocess_bwd_kernel(Batch<PreprocessBwdArgs> batch, int K, int dens_shared) {
    static_assert(VW == 1 || (SUM && VW == BLOCK / 64), "views are split over the waves of a workgroup in SUM mode only");
    // VW == 1: one staging area per wave of the 256-thread workgroup, static.  VW > 1: the workgroup has vw = blockDim.x / 64
    // waves and as many staging areas, dynamic (K = 2 holds 26 KB instead of 51: four workgroups per CU by registers and LDS)
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    GatherLds* const s_gather = VW > 1 ? reinterpret_cast<GatherLds*>(s_dyn) : StaticGather<(VW > 1 ? 0 : BLOCK / 64)>::get();
    const PreprocessBwdArgs& out = batch.v[SUM ? 0 : blockIdx.y];
    const int vw = VW > 1 ? (int)(blockDim.x >> 6) : 1;         // views per workgroup row (VW > 1)
    const int GPB = VW > 1 ? 64 : BLOCK;                        // Gaussians per workgroup
    if ((int)(blockIdx.x * GPB) >= out.P) return;               // workgroup-uniform
    // constant prefix (ExaRasterBackwardJob.grad_first): Gaussians below gf are inputs only -- no chain rule, no output
    // row (row = idx - gf).  A workgroup of constants leaves at once; in the boundary workgroup they stay in the wave
    // for the cooperative gather below but are neither `vis` nor `valid`.
    const int gf = PREFIX ? out.grad_first : 0;
    if (PREFIX && (int)((blockIdx.x + 1) * GPB) <= gf) return;  // workgroup-uniform
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef EXA_PROBE_PBWD      // probe build only (tools/gpu_pbwd_phases.py): phases of every wave on the 100 MHz clock
    const unsigned long long pb_t0 = wall_clock64();
    float pb_ph[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#define PBWD_PHASE(i) pb_ph[i] = (float)(wall_clock64() - pb_t0)
#else
#define PBWD_PHASE(i) do { } while (0)
#endif
    const int idx = VW == 1 ? blockIdx.x * BLOCK + threadIdx.x : blockIdx.x * 64 + lane;
    // NO per-lane early exit: the wave-cooperative gather below needs all 64 lanes, also in the last, partly filled
    // wave (Gaussians appended by densification sit exactly there)
    const bool valid = idx < out.P && idx >= gf;
    const int idc = idx < out.P ? idx : out.P - 1;              // clamped index for loads
    const int row = idx - gf;                                   // output row
    float in_s[3] = {0.f, 0.f, 0.f};
    float4 in_q = make_float4(1.f, 0.f, 0.f, 0.f);
    float in_cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (out.cov3D_precomp) {
#pragma unroll
        for (int i = 0; i < 6; ++i) in_cov[i] = out.cov3D_precomp[idc * 6 + i];
    } else {
        in_s[0] = out.scales[idc * 3 + 0]; in_s[1] = out.scales[idc * 3 + 1]; in_s[2] = out.scales[idc * 3 + 2];
        in_q = reinterpret_cast<const float4*>(out.rotations)[idc];
    }
    const float x = out.means3D[idc * 3 + 0], y = out.means3D[idc * 3 + 1], z = out.means3D[idc * 3 + 2];

    float dmean[3] = {0.f, 0.f, 0.f}, dscale[3] = {0.f, 0.f, 0.f};
    float dq[4] = {0.f, 0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dop = 0.f, dcol[3] = {0.f, 0.f, 0.f};
    bool sh_init = out.accumulate != 0;                         // dL_dsh of this Gaussian already holds an earlier view's values
    //                                                             (or another render's: ExaRasterBackwardJob.accumulate)
    float dn_acc = 0.f, dn_cnt = 0.f, dn_rmax = 0.f;            // dens_shared: this thread's share of the K views' statistics

    // VW == 1: this thread walks views 0 .. K - 1 (SUM) or its own job; VW > 1: this wave's ONE view (none past K)
    const int view_first = VW > 1 ? (int)blockIdx.y * vw + wave : 0;
    const int view_last = VW > 1 ? min(K, view_first + 1) : (SUM ? K : 1);
    // (VW > 1: at most ONE trip, spelled so that the compiler sees it -- the sums below then are this view's values, not
    //  loop-carried accumulators: 198 -> ~120 VGPRs, three waves per SIMD instead of two)
    for (int view = view_first, trip = 0; view < view_last && (VW == 1 || trip == 0); ++view, ++trip) {
        const PreprocessBwdArgs& a = batch.v[SUM ? view : blockIdx.y];
        const float* __restrict__ v = a.viewmatrix;
        const float* __restrict__ p = a.projmatrix;
        // r3 first: the partial gather depends on it
        const uint4 r3 = reinterpret_cast<const uint4*>(a.splats + idc)[3];
        // an overflowed forward left no lists behind: every gradient of that view is zero (the overflow itself is
        // reported to the host through the header, include/exa_raster.h)
        const bool vis = valid && a.header->overflow == 0u && a.radii[idc] > 0;
        const uint8_t* __restrict__ touched = a.touched;
        const float4* __restrict__ prec = a.partials.rec;

        // Large splats (scene Gaussians: hundreds to > 1000 sub-tiles) would turn the per-lane gather below into a
        // serial tail of hundreds of trips in ONE lane: from COOP_MIN instances on, the whole wave fetches that
        // Gaussian's partials together -- lane l takes instances l, l + 64, ... -- and the ten sums are reduced
        // across the wave (c5: 286 -> 148 us; no such splat exists in C3).
        constexpr uint32_t COOP_MIN = 64;
        float co[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        {
            unsigned long long heavy = __ballot(vis && r3.z >= COOP_MIN);
            if (__builtin_expect(heavy != 0ull, 0)) {
                while (heavy) {                                         // wave-uniform loop, all 64 lanes take part
                    const int L = __ffsll((long long)heavy) - 1;
                    heavy &= heavy - 1ull;
                    const uint32_t off = (uint32_t)__shfl((int)r3.w, L, 64), nn = (uint32_t)__shfl((int)r3.z, L, 64);
                    float acc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    for (uint32_t j = (uint32_t)lane; j < nn; j += 64) {
                        if (touched[off + j]) {
                            float4 q0, q1, q2;
                            partial_load(prec, off + j, q0, q1, q2);
                            acc[0] += q0.x; acc[1] += q0.y; acc[2] += q0.z; acc[3] += q0.w;
                            acc[4] += q1.x; acc[5] += q1.y; acc[6] += q1.z; acc[7] += q1.w;
                            acc[8] += q2.x; acc[9] += q2.y;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 10; ++k) {
#pragma unroll
                        for (int d = 32; d > 0; d >>= 1) acc[k] += __shfl_xor(acc[k], d, 64);
                        if (lane == L) co[k] = acc[k];
                    }
                }
            }
        }

        PBWD_PHASE(0);                                          // splat row 3 here (first trip), heavy splats gathered
        // ---- this wave's blended instances (contiguous slots, each written at most once), gathered together ----------
        float own[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        stream_gather(r3.w, (vis && r3.z < COOP_MIN) ? r3.z : 0u, touched, prec, a.partial_bytes, own, s_gather[threadIdx.x >> 6], lane);
        float vmean[3] = {0.f, 0.f, 0.f}, vm2[2] = {0.f, 0.f}, vscale[3] = {0.f, 0.f, 0.f};
        float vq[4] = {0.f, 0.f, 0.f, 0.f}, vcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float vop = 0.f, vcol[3] = {0.f, 0.f, 0.f};
        if (vis) {
            // Contraction off for the whole chain rule: it is compiled into six instantiations of this kernel, the compiler
            // contracts each on its own, and a view's gradients have to come out of the batched (SUM) instantiations as they
            // come out of the single-view ones (dL/dmean2D bit for bit: tests/test_gpu_parity.py, batched views)
#pragma clang fp contract(off)
            float mx = own[0], my = own[1], mxx = own[2], mxy = own[3], myy = own[4], dz_view = own[9];
            vop = own[5]; vcol[0] = own[6]; vcol[1] = own[7]; vcol[2] = own[8];

            PBWD_PHASE(1);                                      // own partial records gathered
            mx += co[0]; my += co[1]; mxx += co[2]; mxy += co[3]; myy += co[4];
            vop += co[5]; vcol[0] += co[6]; vcol[1] += co[7]; vcol[2] += co[8]; dz_view += co[9];

        // ---- recompute the forward quantities ---------------------------------------------------
        const float pvx = ((v[0] * x + v[4] * y) + v[8] * z) + v[12];
        const float pvy = ((v[1] * x + v[5] * y) + v[9] * z) + v[13];
        const float pvz = ((v[2] * x + v[6] * y) + v[10] * z) + v[14];
        const float hx = ((p[0] * x + p[4] * y) + p[8] * z) + p[12];
        const float hy = ((p[1] * x + p[5] * y) + p[9] * z) + p[13];
        const float hw = ((p[3] * x + p[7] * y) + p[11] * z) + p[15];
        const float pw = 1.0f / (hw + 1e-7f);

        float S00, S01, S02, S11, S12, S22;
        float R[9], sc[3];
        float qr = 1.f, qx = 0.f, qy = 0.f, qz = 0.f;
        if (a.cov3D_precomp) {
            S00 = in_cov[0]; S01 = in_cov[1]; S02 = in_cov[2]; S11 = in_cov[3]; S12 = in_cov[4]; S22 = in_cov[5];
        } else {
            sc[0] = a.scale_modifier * in_s[0];
            sc[1] = a.scale_modifier * in_s[1];
            sc[2] = a.scale_modifier * in_s[2];
            const float4 q = in_q;
            qr = q.x; qx = q.y; qy = q.z; qz = q.w;
            R[0] = 1.0f - 2.0f * (qy * qy + qz * qz); R[1] = 2.0f * (qx * qy - qr * qz); R[2] = 2.0f * (qx * qz + qr * qy);
            R[3] = 2.0f * (qx * qy + qr * qz); R[4] = 1.0f - 2.0f * (qx * qx + qz * qz); R[5] = 2.0f * (qy * qz - qr * qx);
            R[6] = 2.0f * (qx * qz - qr * qy); R[7] = 2.0f * (qy * qz + qr * qx); R[8] = 1.0f - 2.0f * (qx * qx + qy * qy);
            float M[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) M[i * 3 + j] = R[i * 3 + j] * sc[j];
            S00 = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
            S01 = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
            S02 = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
            S11 = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
            S12 = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
            S22 = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
        }
        const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
        const float tz = pvz;
        const float txtz = pvx / tz, tytz = pvy / tz;
        const bool clamp_x = txtz < -limx || txtz > limx;
        const bool clamp_y = tytz < -limy || tytz > limy;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
        const float itz = 1.0f / tz, itz2 = itz * itz;
        const float J00 = a.focal_x * itz, J02 = -(a.focal_x * tx) * itz2;
        const float J11 = a.focal_y * itz, J12 = -(a.focal_y * ty) * itz2;
        // Rv[i][j] = v[j*4+i]
        const float T0[3] = {J00 * v[0] + J02 * v[2], J00 * v[4] + J02 * v[6], J00 * v[8] + J02 * v[10]};
        const float T1[3] = {J11 * v[1] + J12 * v[2], J11 * v[5] + J12 * v[6], J11 * v[9] + J12 * v[10]};
        const Cov2DGrad cg = cov2d_conic_grad(S00, S01, S02, S11, S12, S22, T0, T1, mx, my, mxx, mxy, myy);
        const float* ST0 = cg.ST0;
        const float* ST1 = cg.ST1;
        const float dpx = cg.dpx, dpy = cg.dpy, da = cg.da, db = cg.db, dc = cg.dc;

        // ---- cov2D -> Sigma3 (full, unsymmetrised gradient G) and -> T ----------------------------
        // G_jk = da T0j T0k + db T0j T1k + dc T1j T1k
        float G[9];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) G[j * 3 + k] = da * T0[j] * T0[k] + db * T0[j] * T1[k] + dc * T1[j] * T1[k];
        float dT0[3], dT1[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            dT0[j] = 2.0f * da * ST0[j] + db * ST1[j];
            dT1[j] = 2.0f * dc * ST1[j] + db * ST0[j];
        }
        if (a.cov3D_precomp) {
            vcov[0] = G[0]; vcov[1] = G[1] + G[3]; vcov[2] = G[2] + G[6];
            vcov[3] = G[4]; vcov[4] = G[5] + G[7]; vcov[5] = G[8];
        } else {
            // Sigma = M M^T, M = R diag(s):  dL/dM = (G + G^T) M
            float M[9], dM[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) M[i * 3 + j] = R[i * 3 + j] * sc[j];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc += (G[i * 3 + k] + G[k * 3 + i]) * M[k * 3 + j];
                    dM[i * 3 + j] = acc;
                }
            float dR[9];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                vscale[j] = a.scale_modifier * (dM[0 * 3 + j] * R[0 * 3 + j] + dM[1 * 3 + j] * R[1 * 3 + j] + dM[2 * 3 + j] * R[2 * 3 + j]);
#pragma unroll
                for (int i = 0; i < 3; ++i) dR[i * 3 + j] = dM[i * 3 + j] * sc[j];
            }
            vq[0] = 2.0f * (-qz * dR[1] + qy * dR[2] + qz * dR[3] - qx * dR[5] - qy * dR[6] + qx * dR[7]);
            vq[1] = 2.0f * (qy * dR[1] + qz * dR[2] + qy * dR[3] - 2.0f * qx * dR[4] - qr * dR[5] + qz * dR[6] + qr * dR[7] - 2.0f * qx * dR[8]);
            vq[2] = 2.0f * (-2.0f * qy * dR[0] + qx * dR[1] + qr * dR[2] + qx * dR[3] + qz * dR[5] - qr * dR[6] + qz * dR[7] - 2.0f * qy * dR[8]);
            vq[3] = 2.0f * (-2.0f * qz * dR[0] - qr * dR[1] + qx * dR[2] + qr * dR[3] - 2.0f * qz * dR[4] + qy * dR[5] + qx * dR[6] + qy * dR[7]);
        }

        // ---- T = J Rv -> J -> view-space t ---------------------------------------------------------
        const float dJ00 = dT0[0] * v[0] + dT0[1] * v[4] + dT0[2] * v[8];
        const float dJ02 = dT0[0] * v[2] + dT0[1] * v[6] + dT0[2] * v[10];
        const float dJ11 = dT1[0] * v[1] + dT1[1] * v[5] + dT1[2] * v[9];
        const float dJ12 = dT1[0] * v[2] + dT1[1] * v[6] + dT1[2] * v[10];
        const float itz3 = itz2 * itz;
        const float dtx = clamp_x ? 0.f : -a.focal_x * itz2 * dJ02;
        const float dty = clamp_y ? 0.f : -a.focal_y * itz2 * dJ12;
        const float dtz = -a.focal_x * itz2 * dJ00 - a.focal_y * itz2 * dJ11 +
                          2.0f * a.focal_x * tx * itz3 * dJ02 + 2.0f * a.focal_y * ty * itz3 * dJ12 + dz_view;
        // t = [mu, 1] @ viewmatrix[:, :3]
        vmean[0] = dtx * v[0] + dty * v[1] + dtz * v[2];
        vmean[1] = dtx * v[4] + dty * v[5] + dtz * v[6];
        vmean[2] = dtx * v[8] + dty * v[9] + dtz * v[10];

        // ---- pixel centre -> NDC -> clip -> mean ---------------------------------------------------
        vm2[0] = dpx * 0.5f * a.grid.W;
        vm2[1] = dpy * 0.5f * a.grid.H;
        const float dhx = vm2[0] * pw, dhy = vm2[1] * pw;
        const float dhw = -(vm2[0] * hx + vm2[1] * hy) * pw * pw;
        vmean[0] += dhx * p[0] + dhy * p[1] + dhw * p[3];
        vmean[1] += dhx * p[4] + dhy * p[5] + dhw * p[7];
        vmean[2] += dhx * p[8] + dhy * p[9] + dhw * p[11];

        // ---- SH colour ---------------------------------------------------------------------------
        if (SH && a.shs) {
            const uint32_t flags = reinterpret_cast<const uint4*>(a.splats + idc)[0].z;
            const float g[3] = {(flags & 1u) ? 0.f : vcol[0], (flags & 2u) ? 0.f : vcol[1], (flags & 4u) ? 0.f : vcol[2]};
            const float* cp = a.campos;
            const float ux = x - cp[0], uy = y - cp[1], uz = z - cp[2];
            const float inv = 1.0f / sqrtf(ux * ux + uy * uy + uz * uz);
            const float X = ux * inv, Y = uy * inv, Z = uz * inv;
            const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
            const float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                        C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
            const float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                        C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                        C3_6 = -0.5900435899266435f;
            const int deg = a.sh_degree;
            const float xx = X * X, yy = Y * Y, zz = Z * Z, xy = X * Y, yz = Y * Z, xz = X * Z;
            float basis[16], bdx[16], bdy[16], bdz[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { basis[i] = 0.f; bdx[i] = 0.f; bdy[i] = 0.f; bdz[i] = 0.f; }
            basis[0] = C0;
            if (deg > 0) {
                basis[1] = -C1 * Y; bdy[1] = -C1;
                basis[2] = C1 * Z;  bdz[2] = C1;
                basis[3] = -C1 * X; bdx[3] = -C1;
                if (deg > 1) {
                    basis[4] = C2_0 * xy; bdx[4] = C2_0 * Y; bdy[4] = C2_0 * X;
                    basis[5] = C2_1 * yz; bdy[5] = C2_1 * Z; bdz[5] = C2_1 * Y;
                    basis[6] = C2_2 * (2.0f * zz - xx - yy); bdx[6] = -2.0f * C2_2 * X; bdy[6] = -2.0f * C2_2 * Y; bdz[6] = 4.0f * C2_2 * Z;
                    basis[7] = C2_3 * xz; bdx[7] = C2_3 * Z; bdz[7] = C2_3 * X;
                    basis[8] = C2_4 * (xx - yy); bdx[8] = 2.0f * C2_4 * X; bdy[8] = -2.0f * C2_4 * Y;
                    if (deg > 2) {
                        basis[9] = C3_0 * Y * (3.0f * xx - yy); bdx[9] = C3_0 * 6.0f * xy; bdy[9] = C3_0 * (3.0f * xx - 3.0f * yy);
                        basis[10] = C3_1 * xy * Z; bdx[10] = C3_1 * yz; bdy[10] = C3_1 * xz; bdz[10] = C3_1 * xy;
                        basis[11] = C3_2 * Y * (4.0f * zz - xx - yy); bdx[11] = C3_2 * (-2.0f * xy); bdy[11] = C3_2 * (4.0f * zz - xx - 3.0f * yy); bdz[11] = C3_2 * 8.0f * yz;
                        basis[12] = C3_3 * Z * (2.0f * zz - 3.0f * xx - 3.0f * yy); bdx[12] = C3_3 * (-6.0f * xz); bdy[12] = C3_3 * (-6.0f * yz); bdz[12] = C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
                        basis[13] = C3_4 * X * (4.0f * zz - xx - yy); bdx[13] = C3_4 * (4.0f * zz - 3.0f * xx - yy); bdy[13] = C3_4 * (-2.0f * xy); bdz[13] = C3_4 * 8.0f * xz;
                        basis[14] = C3_5 * Z * (xx - yy); bdx[14] = C3_5 * 2.0f * xz; bdy[14] = C3_5 * (-2.0f * yz); bdz[14] = C3_5 * (xx - yy);
                        basis[15] = C3_6 * X * (xx - 3.0f * yy); bdx[15] = C3_6 * (3.0f * xx - 3.0f * yy); bdy[15] = C3_6 * (-6.0f * xy);
                    }
                }
            }
            const int ncoef = (deg + 1) * (deg + 1);
            const float* sh = a.shs + (size_t)idc * a.sh_M * 3;
            float* dsh = (out.dL_dsh && valid) ? out.dL_dsh + (size_t)row * a.sh_M * 3 : nullptr;
            float ddirx = 0.f, ddiry = 0.f, ddirz = 0.f;
            for (int k = 0; k < a.sh_M; ++k) {
                const bool on = k < ncoef && k < 16;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (dsh && valid) {
                        const float val = on ? basis[k] * g[c] : 0.f;
                        dsh[k * 3 + c] = sh_init ? dsh[k * 3 + c] + val : val;
                    }
                    if (on) {
                        const float w = sh[k * 3 + c] * g[c];
                        ddirx += bdx[k] * w; ddiry += bdy[k] * w; ddirz += bdz[k] * w;
                    }
                }
            }
            // dir = u / |u|
            const float dot = X * ddirx + Y * ddiry + Z * ddirz;
            vmean[0] += (ddirx - X * dot) * inv;
            vmean[1] += (ddiry - Y * dot) * inv;
            vmean[2] += (ddirz - Z * dot) * inv;
            sh_init = true;
        }
        }
        PBWD_PHASE(2);                                          // chain rule
        if (valid && a.dL_dmeans2D) {
            a.dL_dmeans2D[row * 3 + 0] = vm2[0]; a.dL_dmeans2D[row * 3 + 1] = vm2[1]; a.dL_dmeans2D[row * 3 + 2] = 0.f;
        }
        // fused densification statistics of this view (reference avatar/main/model.py:279-285 + module.py:155-157): the
        // screen-space gradient is in registers here, no separate pass over the Gaussians
        if (vis) {
            if (SUM && dens_shared) {
                dn_acc += grad_norm2(vm2[0], vm2[1]);
                dn_cnt += 1.0f;
                dn_rmax = fmaxf(dn_rmax, (float)a.radii[idc]);
            } else {
                if (a.dens_accum) a.dens_accum[row] += grad_norm2(vm2[0], vm2[1]);
                if (a.dens_cnt) a.dens_cnt[row] += 1.0f;
                if (a.dens_rmax) a.dens_rmax[row] = fmaxf(a.dens_rmax[row], (float)a.radii[idc]);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) { dmean[i] += vmean[i]; dscale[i] += vscale[i]; dcol[i] += vcol[i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) dq[i] += vq[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) dcov[i] += vcov[i];
        dop += vop;
    }
    // where this workgroup row's sum goes: row 0 writes the outputs, a further row (views 4 ..) a scratch that sum_groups_kernel adds
    float* o_mean = out.dL_dmeans3D; float* o_op = out.dL_dopacity; float* o_col = out.dL_dcolors; float* o_scale = out.dL_dscales;
    float* o_q = out.dL_drotations; float* o_cov = out.dL_dcov3D;
    float* o_dens = nullptr;                                    // (scratch rows only: acc, cnt, rmax of this row's views)
    if (VW > 1) {                                               // add the views of the workgroup's waves
        wave_lds_fence();
        __syncthreads();                                        // every wave is done with its staging area: reuse it
        float* s_red = reinterpret_cast<float*>(s_dyn);
        const int B = vw * 64;
        float* r = s_red + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 3; ++i) { r[i * B] = dmean[i]; r[(3 + i) * B] = dscale[i]; r[(17 + i) * B] = dcol[i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) r[(6 + i) * B] = dq[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) r[(10 + i) * B] = dcov[i];
        r[16 * B] = dop;
        r[20 * B] = dn_acc; r[21 * B] = dn_cnt; r[22 * B] = dn_rmax;
        __syncthreads();
        if (wave != 0) return;
        auto total = [&](int f) {                               // (the association of the four-wave workgroup of rounds 2-5)
            const float* q = s_red + f * B + lane;
            return vw == 4 ? (q[0] + q[64]) + (q[128] + q[192]) : vw == 3 ? (q[0] + q[64]) + q[128] : vw == 2 ? q[0] + q[64] : q[0];
        };
#pragma unroll
        for (int i = 0; i < 3; ++i) { dmean[i] = total(i); dscale[i] = total(3 + i); dcol[i] = total(17 + i); }
#pragma unroll
        for (int i = 0; i < 4; ++i) dq[i] = total(6 + i);
#pragma unroll
        for (int i = 0; i < 6; ++i) dcov[i] = total(10 + i);
        dop = total(16);
        dn_acc = total(20); dn_cnt = total(21);
        {
            const float* q = s_red + 22 * B + lane;
            dn_rmax = q[0];
            for (int w = 1; w < vw; ++w) dn_rmax = fmaxf(dn_rmax, q[64 * w]);
        }
        if (blockIdx.y > 0) {
            float* sc = batch.v[blockIdx.y * vw].group_scratch;         // SoA, field f of Gaussian row at sc[f * P + row]
            const size_t Pn = (size_t)out.P;
            o_mean = out.dL_dmeans3D ? sc : nullptr; o_scale = out.dL_dscales ? sc + 3 * Pn : nullptr;
            o_q = out.dL_drotations ? sc + 6 * Pn : nullptr; o_cov = out.dL_dcov3D ? sc + 10 * Pn : nullptr;
            o_op = out.dL_dopacity ? sc + 16 * Pn : nullptr; o_col = out.dL_dcolors ? sc + 17 * Pn : nullptr;
            o_dens = sc + 20 * Pn;
        }
    }
    if (!valid) return;
    if (o_dens) {                                               // a further workgroup row: everything into its scratch rows
        o_dens[3 * row + 0] = dn_acc; o_dens[3 * row + 1] = dn_cnt; o_dens[3 * row + 2] = dn_rmax;
    } else if (SUM && dens_shared && dn_cnt > 0.f) {            // one read-modify-write per Gaussian for all views of this row
        if (out.dens_accum) out.dens_accum[row] += dn_acc;
        if (out.dens_cnt) out.dens_cnt[row] += dn_cnt;
        if (out.dens_rmax) out.dens_rmax[row] = fmaxf(out.dens_rmax[row], dn_rmax);
    }
    if (SH && !sh_init && out.shs && out.dL_dsh) {                    // never visible: the SH block above did not run
        float* dsh = out.dL_dsh + (size_t)row * out.sh_M * 3;
        for (int k = 0; k < out.sh_M * 3; ++k) dsh[k] = 0.f;
    }

    if (out.accumulate) {                                       // ExaRasterBackwardJob.accumulate: out = held + this render's
        if (out.dL_dmeans3D) { dmean[0] += out.dL_dmeans3D[row * 3 + 0]; dmean[1] += out.dL_dmeans3D[row * 3 + 1]; dmean[2] += out.dL_dmeans3D[row * 3 + 2]; }
        if (out.dL_dopacity) dop += out.dL_dopacity[row];
        if (out.dL_dcolors) { dcol[0] += out.dL_dcolors[row * 3 + 0]; dcol[1] += out.dL_dcolors[row * 3 + 1]; dcol[2] += out.dL_dcolors[row * 3 + 2]; }
        if (out.dL_dscales) { dscale[0] += out.dL_dscales[row * 3 + 0]; dscale[1] += out.dL_dscales[row * 3 + 1]; dscale[2] += out.dL_dscales[row * 3 + 2]; }
        if (out.dL_drotations) {
            const float4 h = reinterpret_cast<const float4*>(out.dL_drotations)[row];
            dq[0] += h.x; dq[1] += h.y; dq[2] += h.z; dq[3] += h.w;
        }
        if (out.dL_dcov3D) {
#pragma unroll
            for (int i = 0; i < 6; ++i) dcov[i] += out.dL_dcov3D[row * 6 + i];
        }
    }
    if (o_mean) { o_mean[row * 3 + 0] = dmean[0]; o_mean[row * 3 + 1] = dmean[1]; o_mean[row * 3 + 2] = dmean[2]; }
    if (o_op) o_op[row] = dop;
    if (o_col) { o_col[row * 3 + 0] = dcol[0]; o_col[row * 3 + 1] = dcol[1]; o_col[row * 3 + 2] = dcol[2]; }
    if (o_scale) { o_scale[row * 3 + 0] = dscale[0]; o_scale[row * 3 + 1] = dscale[1]; o_scale[row * 3 + 2] = dscale[2]; }
    if (o_q) reinterpret_cast<float4*>(o_q)[row] = make_float4(dq[0], dq[1], dq[2], dq[3]);
    if (o_cov) {
#pragma unroll
        for (int i = 0; i < 6; ++i) o_cov[row * 6 + i] = dcov[i];
    }
#ifdef EXA_PROBE_PBWD
    PBWD_PHASE(3);
    {   // the wave's phases (max over its lanes), as floats in the z slots of its own rows of dL_dmeans2D
        float m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            m[i] = pb_ph[i];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) m[i] = fmaxf(m[i], __shfl_xor(m[i], d, 64));
        }
        if (lane < 4 && out.dL_dmeans2D) out.dL_dmeans2D[(row - lane + lane) * 3 + 2] = lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : m[3];
        if (lane == 4 && out.dL_dmeans2D) out.dL_dmeans2D[row * 3 + 2] = (float)(pb_t0 & 0xffffff);
    }
#endif
}

// Fused densification statistics (include/exa_raster.h: exa_raster_densify_stats; reference
// avatar/main/model.py:279-285 + avatar/common/nets/module.py:155-157).  Pure streaming: 16 B in, <= 12 B
// read-modify-write per Gaussian.
__global__ __launch_bounds__(BLOCK) void densify_stats_kernel(int P, const float* __restrict__ g2d,
                                                              const int32_t* __restrict__ radii, float* accum,
                                                              float* cnt, float* rmax) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    if (accum) {
        const float gx = g2d[3 * i], gy = g2d[3 * i + 1];
        accum[i] += grad_norm2(gx, gy);
    }
    if (cnt) cnt[i] += 1.0f;
    if (rmax) rmax[i] = fmaxf(rmax[i], (float)r);
}
hipError_t launch_densify_stats(int P, const float* g2d, const int32_t* radii, float* accum, float* cnt, float* rmax,
                                hipStream_t s) {
    if (P == 0) return hipSuccess;
    densify_stats_kernel<<<(P + BLOCK - 1) / BLOCK, BLOCK, 0, s>>>(P, g2d, radii, accum, cnt, rmax);
    return hipGetLastError();
}

// Second pass of a summed batch of more than four views (preprocess_bwd_kernel<SUM, VW = 4>): the sums of the further
// workgroup rows (views 4 ..), left in their scratch in the layout of the outputs, are added to the outputs of row 0 in row
// order.  68 B read + 68 B read-modify-write per Gaussian.
__global__ __launch_bounds__(BLOCK) void sum_groups_kernel(PreprocessBwdArgs out, const float* __restrict__ sc, int dens_shared) {
    const int row = blockIdx.x * BLOCK + threadIdx.x;
    if (row >= out.P) return;
    const size_t Pn = (size_t)out.P;
    auto add = [&](float* dst, const float* src, int n) {
        if (!dst) return;
        for (int i = 0; i < n; ++i) dst[(size_t)row * n + i] += src[(size_t)row * n + i];
    };
    add(out.dL_dmeans3D, sc, 3); add(out.dL_dscales, sc + 3 * Pn, 3); add(out.dL_drotations, sc + 6 * Pn, 4);
    add(out.dL_dcov3D, sc + 10 * Pn, 6); add(out.dL_dopacity, sc + 16 * Pn, 1); add(out.dL_dcolors, sc + 17 * Pn, 3);
    const float* dn = sc + 20 * Pn + 3 * (size_t)row;
    if (dens_shared && dn[1] > 0.f) {
        if (out.dens_accum) out.dens_accum[row] += dn[0];
        if (out.dens_cnt) out.dens_cnt[row] += dn[1];
        if (out.dens_rmax) out.dens_rmax[row] = fmaxf(out.dens_rmax[row], dn[2]);
    }
}

hipError_t launch_preprocess_bwd(const PreprocessBwdArgs* a, int K, int sum_shared, int dens_shared, hipStream_t s) {
    int P = 0;
    for (int k = 0; k < K; ++k) P = max(P, a[k].P);
    if (P == 0) return hipSuccess;
    bool sh = false;
    bool prefix = false;
    for (int k = 0; k < K; ++k) { sh = sh || a[k].shs != nullptr; prefix = prefix || a[k].grad_first > 0; }
    const dim3 grid256((P + BLOCK - 1) / BLOCK, sum_shared ? 1 : K);
    if (sum_shared && !sh) {
        // one wave per view: workgroups of vw = min(K, 4) waves over 64 Gaussians, ceil(K / vw) rows of them
        const int vw = K < BLOCK / 64 ? K : BLOCK / 64, rows = (K + vw - 1) / vw;
        for (int r = 1; r < rows; ++r)
            if (!a[r * vw].group_scratch) return hipErrorInvalidValue;
        preprocess_bwd_kernel<true, BLOCK / 64, false, false><<<dim3((P + 63) / 64, rows), 64 * vw, (size_t)vw * sizeof(GatherLds), s>>>(
            make_batch(a, K), K, dens_shared);
        for (int r = 1; r < rows; ++r)
            sum_groups_kernel<<<(P + BLOCK - 1) / BLOCK, BLOCK, 0, s>>>(a[0], a[r * vw].group_scratch, dens_shared);
    } else if (sum_shared)
        preprocess_bwd_kernel<true, 1, true, false><<<grid256, BLOCK, 0, s>>>(make_batch(a, K), K, dens_shared);
    else if (sh && prefix)
        preprocess_bwd_kernel<false, 1, true, true><<<grid256, BLOCK, 0, s>>>(make_batch(a, K), K, dens_shared);
    else if (sh)
        preprocess_bwd_kernel<false, 1, true, false><<<grid256, BLOCK, 0, s>>>(make_batch(a, K), K, dens_shared);
    else if (prefix)
        preprocess_bwd_kernel<false, 1, false, true><<<grid256, BLOCK, 0, s>>>(make_batch(a, K), K, dens_shared);
    else
        preprocess_bwd_kernel<false, 1, false, false><<<grid256, BLOCK, 0, s>>>(make_batch(a, K), K, dens_shared);
    return hipGetLastError();
}

}  // namespace exa
