// Per-batch backward of the alpha compositing for gfx950 -- atomic-free, one wave per 64-entry batch slot.
//
// Unit of work = a BATCH SLOT: 64 consecutive entries of one sub-tile's sorted list x the 64 pixels of that
// sub-tile.  The forward pass checkpoints the per-pixel state (T, C_rgb, depth; stopped pixels as -T) at the
// start of every batch and at its exit, so every batch of every sub-tile runs concurrently and the longest
// dependent chain is 64 splats (a kernel that walks whole lists with one wave is bound by its longest list).
//
// COMPACTION.  The forward pass also leaves one 64-bit mask per batch: which entries it blended into at least one
// pixel.  An entry that touched no live pixel is the identity of the recurrence (alpha_eff = 0 everywhere) and gets
// no gradient, so the backward pass drops it before doing anything else: the flagged entries are staged densely in
// LDS (rank = mbcnt of the mask) and everything below runs on that shorter list.  On avatar-like scenes (opaque,
// small splats) only ~60 % of the entries of the batches the forward enters survive, ~46 % of all instances
// (C3, ring view 0: 893 k instances, 690 k in entered batches, 409 k blended).  Skip decisions depend on alpha only and
// stay bit-identical to the forward's, and so does the transmittance: the recurrence T_{j+1} = T_j - alpha_j T_j (blend.h)
// leaves T untouched by a skipped entry, so the replay on the compacted list walks through exactly the forward's values
// (rounds 2-5 multiplied partial products in groups of four, which the compaction regrouped: an ulp apart).
//
// The kernel is bound by VALU issue (profiles/r04_pmc.md: the VALU pipes are busy 81 % of the launch incl. its ramp and tail,
// five waves per SIMD at 83 VGPRs / 8 128 B of LDS per wave; 13-16 % of the HBM roofline), so the design minimises instructions
// per (pixel, splat) pair.  Two phases per chunk of GC = 8 splats (GC = 16: the round-2 layout, EXA_BWD_GC=16):
//
//  Phase A  (lane = pixel, splats in list order): REPLAY the forward recurrence from the checkpoint with the
//           forward's own code (blend.h) -- no 1/(1-alpha) reconstruction of T, no `n_contrib` array -- and
//           get dL/d(alpha_i) from a running scalar instead of per-channel suffix colours:
//               R_i = (C_fin - C_i) . g - tail,      tail = T_fin (g_alpha - bg . g)
//               dL/d(alpha_i) = T_i (c_i . g) - R_{i+1} / (1 - alpha_i)
//           ("." also runs over the depth channel).  Two numbers per pair go to LDS as one 8-byte store:
//           aG = (opacity G) dL/dalpha and the blend weight w = alpha T.  ~160 VALU instructions per chunk (236 in round 5:
//           blend.h's two-instruction recurrence, Horner power, gated product instead of a `w > 0` select).
//  Phase B  (lane = (splat g, pixel row h), XLayout<8>): the transposed read (ds_read_b128, entry stride 132 floats:
//           conflict-free in gfx950's lane groups) turns the per-splat sums over the 64 pixels into per-lane accumulation.
//           The screen-space moments are accumulated against COMPILE-TIME pixel coordinates (x - 4 in -4..3):
//           sum aG, sum aG x, sum aG x^2 per pixel row, shifted to the splat centre once per chunk; + 3 (4) FMAs for the
//           colour (depth) sums.  The eight pixel rows of a splat sit in eight consecutive lanes and are combined with
//           three v_add_f32_dpp each (one asm block for the nine sums).  92 VALU instructions per chunk.
//  All staged float4 reads stay 16 bytes wide (keep_b128): narrowed to ds_read_b96 they cost 3.6 M bank-conflict cycles
//  per dispatch in round 3.
//
// Only blended instances get their 40-byte Gaussian-major partial record written (exactly once: no memset, NO atomic
// in the whole backward pass -- device-scope fp32 atomics run at ~12 G/s on MI355X -- bit-deterministic) and their
// `touched` byte set; preprocess_bwd.hip reads the bytes and fetches only those records.
//
// Replaces upstream BACKWARD::renderCUDA of the rasterizer behind reference
// avatar/common/nets/module.py:632-640 (backward reached from avatar/main/train.py:46).
// Derivatives follow oracle/raster_oracle.py (autograd of oracle step 9/10) with straight-through
// min(0.99, .).  The screen-space gradients are emitted as five moments of s = dL/dG * G
// (sum s dx, s dy, s dx^2, s dx dy, s dy^2); preprocess_bwd.hip turns them into d/d(mean2D, conic).
//
// CONSTANT PREFIX (PREFIX = true; ExaRasterBackwardJob.grad_first > 0): ExAvatar's composite renders blend the detached
// scene under the human (torch.cat((scene.detach(), human)), reference avatar/main/model.py:119-126).  Gaussians below
// `grad_first` get no gradient: a batch whose blended entries are all constants exits after its header (most of the
// image in a scene + human render), inside a live batch a chunk of 16 constants skips phase B, and no partial record is
// written for a constant entry.  The arithmetic of the trainable entries is the same source, but it is compiled as its
// own instantiation: folded into the one kernel -- even behind scalar branches on grad_first -- the extra tests cost
// the all-trainable path 2.5-3 us of 55 on C3 (A/B on one box), so that path carries none of them.  Between the two
// instantiations the compiler may contract multiply-adds differently: the trainable gradients agree to rounding
// (~1e-7 relative), not bit for bit.
//
// Algorithmic HBM bytes: reads 4 B/instance (sorted ids), 64 B per gathered (blended) splat, 12 (+8) B/pixel of
// incoming gradient per batch, 2 x 20 B/pixel of checkpoints per batch; writes 41 B per blended instance.
#include <stdlib.h>
#include "blend.h"

namespace exa {

constexpr int RBLOCK = 64;            // ONE wave per workgroup
// Chunk = GC splats x 64 pixels in the transposition buffer; lane (g, h) of phase B sums splat g over the GC pixels of
// pixel group h (64 / GC groups).  Layouts found by exhaustive search for conflict-free ds_read_b128 rows:
//   GC = 16: row = 16 pixels x {aG, w} + 4 pad = 36 floats, group stride 16 rows            (9 216 B, 3 waves / SIMD)
//   GC =  8: row =  8 pixels x {aG, w}         = 16 floats, group stride 132 floats         (4 224 B, 5 waves / SIMD;
//            one pixel ROW per group: no y moments inside the loop, but three shuffle steps instead of two)
template <int GC> struct XLayout;
template <> struct XLayout<16> {
    static constexpr int ROW = 36, GROUP = 16 * 36, FLOATS = 4 * 16 * 36;
    static constexpr int WK = ROW;                                          // floats between consecutive entries (phase A)
    __device__ static __forceinline__ int g(int lane) { return lane % 16; }      // phase B: entry of the chunk ...
    __device__ static __forceinline__ int h(int lane) { return lane / 16; }      // ... and pixel group (two pixel rows)
    __device__ static __forceinline__ int wbase(int lane) { return (lane / 16) * GROUP + 2 * (lane % 16); }
    __device__ static __forceinline__ int rbase(int lane) { return h(lane) * GROUP + g(lane) * ROW; }
    __device__ static __forceinline__ float reduce(float x) {              // sum over the four pixel groups of an entry
        x += __shfl_xor(x, 16, 64);
        x += __shfl_xor(x, 32, 64);
        return x;
    }
    __device__ static __forceinline__ void reduce9(float& a, float& b, float& c, float& d, float& e, float& f, float& g,
                                                   float& h, float& i) {
        a = reduce(a); b = reduce(b); c = reduce(c); d = reduce(d); e = reduce(e); f = reduce(f); g = reduce(g);
        h = reduce(h); i = reduce(i);
    }
};
// GC = 8, entry-major lanes: lane = g * 8 + h sums pixel ROW h of entry g, so the eight partial sums of an entry sit in
// eight CONSECUTIVE lanes and are combined with three DPP adds (quad_perm xor 1, xor 2, row_half_mirror) -- plain VALU
// instructions, no LDS-pipe shuffle (the first GC = 8 layout was row-major: xor 8 / 16 / 32 = three ds_bpermute rounds on
// ten values per chunk, measured 63.5 vs 56.4 us).  Layout [entry][pixel row][8 pixels x {aG, w}], entry stride 132
// floats: phase-A stores of one entry are 64 contiguous 8-byte words, phase-B ds_read_b128 of the 16 lanes a quarter-wave
// services start at 16 distinct multiples of 4 banks (entry stride = 4 banks mod 16).  4 224 B instead of 9 216:
// 8 128 B of LDS per wave = 5 waves per SIMD instead of 3.
template <> struct XLayout<8> {
    static constexpr int ROW = 16, GROUP = 132, FLOATS = 8 * 132;
    static constexpr int WK = GROUP;
    __device__ static __forceinline__ int g(int lane) { return lane >> 3; }
    __device__ static __forceinline__ int h(int lane) { return lane & 7; }
    __device__ static __forceinline__ int wbase(int lane) { return (lane >> 3) * ROW + 2 * (lane & 7); }
    __device__ static __forceinline__ int rbase(int lane) { return g(lane) * GROUP + h(lane) * ROW; }
    __device__ static __forceinline__ float reduce(float x) {
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));  // row_half_mirror
        return x;
    }
    // The nine (ten) sums of a chunk, three stages of v_add_f32_dpp each.  Written out as ONE asm block because the
    // compiler folds only some of the update_dpp + add pairs of reduce() (round 4 ISA: 21 v_mov_b32_dpp + 21 v_add_f32
    // next to 6 fused v_add_f32_dpp, of 100 VALU instructions per chunk in phase B).  Stage by stage over all values: the
    // DPP read of a register follows its VALU write by N - 1 >= 8 instructions; the s_nop covers the two wait states
    // the first stage needs after the compiler's own writes (its hazard recognizer does not look into asm).
#define EXA_DPP3(r) \
    "v_add_f32_dpp " r ", " r ", " r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define EXA_DPP4(r) \
    "v_add_f32_dpp " r ", " r ", " r " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define EXA_DPP5(r) \
    "v_add_f32_dpp " r ", " r ", " r " row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define EXA_STAGE9(M) M("%0") M("%1") M("%2") M("%3") M("%4") M("%5") M("%6") M("%7") M("%8")
    __device__ static __forceinline__ void reduce9(float& a, float& b, float& c, float& d, float& e, float& f, float& g,
                                                   float& h, float& i) {
        asm("s_nop 1\n" EXA_STAGE9(EXA_DPP3) EXA_STAGE9(EXA_DPP4) EXA_STAGE9(EXA_DPP5)
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i));
    }
#undef EXA_STAGE9
#undef EXA_DPP3
#undef EXA_DPP4
#undef EXA_DPP5
};

// Without the depth channel the fourth component of a staged float4 is dead, and the compiler narrows the LDS read to
// ds_read_b96 -- which gfx950 services in EIGHT lane groups with banks mod 32 (MI355X_MICROARCH.md, LDS table): 8 cycles
// instead of 4 for a broadcast, and the s_pg reads of phase B (rows laid out for ds_read_b128's mod-64 banks) become
// 2-way conflicts in every group, 16 cycles instead of 4 (round 3's PMC pass: 3.6 M SQ_LDS_BANK_CONFLICT cycles per
// dispatch = 64 per chunk = these eight reads).  An empty asm that passes .x through and names .w as an input keeps the
// read 16 bytes wide: no instruction, and -- not being volatile -- no constraint on the scheduling of the reads (a
// volatile one serialised phase B into read / wait / read / wait).
__device__ __forceinline__ void keep_b128(float4& v) { asm("" : "+v"(v.x) : "v"(v.w)); }

__device__ __forceinline__ int mask_rank(uint32_t lo, uint32_t hi) {          // set bits of (hi:lo) below this lane
    return (int)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
}

// What a wave needs to know about a batch slot before it can fetch the batch (round trip 1) ...
struct BwdHdr {
    uint32_t st1, begin, n;          // owner record: sub-tile + 1 (0 = unused / end slot), list begin, list length
    uint32_t bm_lo, bm_hi;           // blended mask
    uint32_t id;                     // this lane's sorted id (garbage past the end of the list; only flagged lanes use theirs)
};
// ... and the batch itself (round trip 2): this lane's splat record, partial slot, the per-pixel state at the start of the
// batch and at the forward's exit, the incoming pixel gradient.
struct BwdPay {
    float2 r0; float4 r1, r2;
    uint32_t pslot;
    float cs0, cs1, cs2, cs3, cs4, cf0, cf1, cf2, cf3, cf4;
    float gr, gg, gb, gd, ga;
};

constexpr uint32_t NO_SLOT = 0xffffffffu;         // s_pslot of a constant (frozen) entry: nothing to write

// WPB: independent waves per workgroup (each with its own slot and its own LDS; no barrier anywhere): a developer knob
// (EXA_BWD_WPB = 2 | 4), default 1.  Measured in round 4 on the suspicion that launches over mostly EMPTY instance buffers
// (the composites' backward: 80 k slots, a tenth of them with work) are bound by the dispatch of their workgroups: they
// are not -- four waves per workgroup cost the five-render iteration 0.909 -> 0.968 ms with the composites alone and
// 1.045 ms with the plain backward too, and the C3 headline 6 415 -> 6 365 it/s (a workgroup holds its LDS and its
// dispatch slot until its slowest wave is done).
template <bool HAS_DEPTH, int GC, int SPW, bool PREFIX, int WPB>
__global__ __launch_bounds__(RBLOCK * WPB) void render_bwd_kernel(Batch<RenderBwdArgs> batch) {
    typedef XLayout<GC> XL;
    __shared__ BatchLds s_b_[WPB];
    __shared__ float4 s_pg_[WPB][4 * 17];       // incoming gradient of each pixel (r, g, b, depth), 16 per pixel group;
                                                // group stride 17: the two groups one ds_read_b128 quarter-wave sees
                                                // sit on different banks
    __shared__ uint32_t s_pslot_[WPB][64];      // Partial slot of each staged splat
    __shared__ __attribute__((aligned(16))) float s_x_[WPB][XLayout<GC>::FLOATS];   // {aG, w}[pixel group][splat][pixel of the group]

    const RenderBwdArgs& a = batch.v[blockIdx.y];
    const uint32_t nslots = (uint32_t)(a.capacity / BATCH);
    const int lane = threadIdx.x & 63;
    const int wv = WPB > 1 ? (int)(threadIdx.x >> 6) : 0;
    const uint32_t wg = (uint32_t)blockIdx.x * WPB + (uint32_t)wv;        // this wave's position in the launch
    BatchLds& s_b = s_b_[wv];
    float4* const s_pg = s_pg_[wv];
    uint32_t* const s_pslot = s_pslot_[wv];
    float* const s_x = s_x_[wv];
#ifdef EXA_PROBE_BWDLINE   // probe build only (tools/gpu_bwd_timeline.py): start / end of every wave on the chip-wide 100 MHz clock
    struct TL { const RenderBwdArgs& a; unsigned long long t0; int lane;
        __device__ ~TL() { if (lane == 0 && 2 * blockIdx.x + 1 < (uint32_t)(a.grid.cells * BIN_PARTS * SUBS_PER_CELL)) {
            a.tw.part_cnt[2 * blockIdx.x] = (uint32_t)t0; a.tw.part_cnt[2 * blockIdx.x + 1] = (uint32_t)wall_clock64(); } } } tl{a, (unsigned long long)wall_clock64(), lane};
#endif
    const float* __restrict__ bg = a.bg;
    // (graph replays with a new gradient tensor per iteration: include/exa_raster.h, dL_dcolor_indirect; a scalar load)
    // (the cast matters: a pointer LOADED from memory is generic to the compiler, and flat loads also count on lgkmcnt --
    //  every LDS / scalar wait behind them would wait for HBM)
    typedef const float __attribute__((address_space(1)))* gfloat_ptr;
    typedef const unsigned long long __attribute__((address_space(4)))* table_ptr;       // constant: a scalar load
    const gfloat_ptr dL_dcolor = a.dL_dcolor_ind ? (gfloat_ptr)(*(table_ptr)(unsigned long long)a.dL_dcolor_ind)
                                                 : (gfloat_ptr)a.dL_dcolor;
    float4* __restrict__ prec = a.partials.rec;
    uint8_t* __restrict__ touched = a.bw.touched;

    auto load_hdr = [&](uint32_t slot) -> BwdHdr {
        BwdHdr h = {0u, 0u, 0u, 0u, 0u, 0u};
        if (slot < nslots) {
            const uint4 own = a.bw.owner[slot];
            const unsigned long long bm = a.bw.bmask[slot];
            h.id = a.bw.sorted[(size_t)slot * BATCH + lane];
            h.st1 = own.x; h.begin = own.y; h.n = own.z;
            h.bm_lo = (uint32_t)bm; h.bm_hi = (uint32_t)(bm >> 32);
        }
        return h;
    };
    const uint32_t grad_first = PREFIX ? (uint32_t)a.grad_first : 0u;
    // composite render (compose.hip): ids of two sources, bit 31 = source B = the trainable Gaussians; A is constant
    const bool two = PREFIX && a.splats2 != nullptr;
    auto trainable = [&](uint32_t id) -> bool { return two ? (id & SRC_B) != 0u : id >= grad_first; };
    auto is_active = [&](const BwdHdr& h) -> bool {              // wave-uniform
        uint32_t lo = __builtin_amdgcn_readfirstlane(h.bm_lo), hi = __builtin_amdgcn_readfirstlane(h.bm_hi);
        if (PREFIX) {                                            // blended AND trainable (lanes past the list end hold
            const unsigned long long tr = __ballot(trainable(h.id));        // garbage ids, but their mask bits are clear)
            lo &= (uint32_t)tr; hi &= (uint32_t)(tr >> 32);
        }
        return __builtin_amdgcn_readfirstlane(h.st1) != 0u && (lo | hi) != 0u;
    };
    // (Round 4 built the alternative -- checkpoints and gradients first, then the splat rows by buffer loads pushed out of
    //  range for lanes without a blended entry, everything in ONE round trip instead of two; commit "render_bwd: one-round-trip
    //  payload loads" -- and measured 6 538 against 6 559 it/s on C3, three interleaved runs each: at five waves per SIMD the
    //  prologue's latency is hidden, and the kernel is VALU-bound.  Not kept.)
    auto load_pay = [&](const BwdHdr& h, uint32_t slot) -> BwdPay {
        BwdPay p;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        p.r0 = make_float2(0.f, 0.f); p.r1 = zero4; p.r2 = zero4; p.pslot = 0u;
        p.cs0 = 1.0f; p.cs1 = p.cs2 = p.cs3 = p.cs4 = 0.f;
        p.cf0 = p.cf1 = p.cf2 = p.cf3 = p.cf4 = 0.f;
        p.gr = p.gg = p.gb = p.gd = p.ga = 0.f;
        if (!is_active(h)) return p;
        const uint32_t bm_lo = __builtin_amdgcn_readfirstlane(h.bm_lo), bm_hi = __builtin_amdgcn_readfirstlane(h.bm_hi);
        const int st = (int)__builtin_amdgcn_readfirstlane(h.st1) - 1;
        const int n = (int)__builtin_amdgcn_readfirstlane(h.n);
        const int b0 = (int)(__builtin_amdgcn_readfirstlane(h.begin) / BATCH);
        const SubTile sub = decode_subtile(st, a.grid);
        const bool flagged = (((lane < 32 ? bm_lo : bm_hi) >> (lane & 31)) & 1u) != 0u;
        if (flagged) {
            const Splat* base = a.splats;
            uint32_t idx = h.id, last = (uint32_t)(a.P - 1);
            if (two) {
                idx = h.id & ~SRC_B;
                if (h.id & SRC_B) { base = a.splats2; last = (uint32_t)(a.P2 - 1); }
            }
            const float4* rec = reinterpret_cast<const float4*>(base + min(idx, last));
            p.r0 = *reinterpret_cast<const float2*>(rec); p.r1 = rec[1]; p.r2 = rec[2];
            const uint4 r3 = reinterpret_cast<const uint4*>(rec)[3];
            const int sx0 = r3.x & 0xffff, sx1 = r3.x >> 16, sy0 = r3.y & 0xffff;
            p.pslot = r3.w + (uint32_t)((sub.gsy - sy0) * (sx1 - sx0) + (sub.gsx - sx0));
            if (PREFIX && !trainable(h.id)) p.pslot = NO_SLOT;
        }
        // state at the START of this batch and at the forward's exit (the sub-tile's end slot)
        const float* cf = a.bw.ckpt + (size_t)(b0 + (n + BATCH - 1) / BATCH) * (5 * 64) + lane;
        const float* cs = a.bw.ckpt + (size_t)slot * (5 * 64) + lane;
        if ((int)slot > b0) { p.cs0 = cs[0]; p.cs1 = cs[64]; p.cs2 = cs[128]; p.cs3 = cs[192]; p.cs4 = HAS_DEPTH ? cs[256] : 0.f; }
        p.cf0 = cf[0]; p.cf1 = cf[64]; p.cf2 = cf[128]; p.cf3 = cf[192]; p.cf4 = HAS_DEPTH ? cf[256] : 0.f;
        const int pxi = sub.ox + (lane & 7), pyi = sub.oy + (lane >> 3);
        if (pxi < a.grid.W && pyi < a.grid.H) {
            const size_t HW = (size_t)a.grid.W * a.grid.H;
            const size_t pix = (size_t)pyi * a.grid.W + pxi;
            p.gr = dL_dcolor[pix];
            p.gg = dL_dcolor[HW + pix];
            p.gb = dL_dcolor[2 * HW + pix];
            if (HAS_DEPTH && a.dL_ddepth) p.gd = a.dL_ddepth[pix];
            if (a.dL_dalpha) p.ga = a.dL_dalpha[pix];
        }
        return p;
    };

    // SPW consecutive batch slots per wave (default 1): the headers of all of them are fetched in one round trip, their
    // payloads in a second one, then the batches are computed one after the other.  Measured on C3: SPW 1 / 2 / 4 =
    // 55.6 / 64.0 / 70.7 us, and persistent waves striding over the slots with a two-deep prefetch 81 us -- the batches'
    // costs differ by an order of magnitude (1..64 blended entries), so anything that takes work distribution away from
    // the hardware's wave dispatcher loses more to imbalance than it gains from hidden latency.
    // Launch order (render_fwd.hip order_slots): wave w takes the w-th batch of the batch-major order the sort launch left
    // in the (dead) bucket array -- heavy batches first, nothing dispatched for slots without work beyond the one load of
    // the waves past the end.  Composites, the EXA_BWD_SPW variants and batched launches (K > 1 jobs: the stream of waves of
    // several jobs balances itself, and the extra dependent load cost 2.5 % there: 8 490 -> 8 260 it/s at K = 8; the three plain
    // renders of the five-render iteration, K = 3: 0.755 -> 0.767 ms per iteration with the order) keep the slot order.
    uint32_t first_slot = wg * SPW;
    if (SPW == 1 && gridDim.y == 1) {
        const uint32_t m0 = a.tw.bwd_meta[0], m1 = a.tw.bwd_meta[1], magic = a.tw.bwd_meta[2];
        // (requested together with the three words above: one trip; a composite's bin workspace has no bucket array)
        const uint32_t mapped = a.bw.bucket ? reinterpret_cast<const uint32_t*>(a.bw.bucket)[min(wg, nslots - 1u)] : 0u;
        if (magic == BWD_ORDER_MAGIC && a.bw.bucket) {
            if (wg >= m0 + m1) return;
            first_slot = mapped;
        }
    }
    BwdHdr hh[SPW];
    BwdPay pp[SPW];
#pragma unroll
    for (int i = 0; i < SPW; ++i) hh[i] = load_hdr(first_slot + i);
#pragma unroll
    for (int i = 0; i < SPW; ++i) pp[i] = load_pay(hh[i], first_slot + i);
#pragma unroll
    for (int rep = 0; rep < SPW; ++rep) {
    const BwdHdr& h0 = hh[rep];
    const BwdPay& p0 = pp[rep];
    const uint32_t slot = first_slot + rep;
    if (is_active(h0)) {
    // ================================= one batch =================================================================
    const uint32_t bm_lo = __builtin_amdgcn_readfirstlane(h0.bm_lo), bm_hi = __builtin_amdgcn_readfirstlane(h0.bm_hi);
    const int st = (int)__builtin_amdgcn_readfirstlane(h0.st1) - 1;
    const int b0 = (int)(__builtin_amdgcn_readfirstlane(h0.begin) / BATCH);   // first slot of the sub-tile
    const int bq = (int)slot - b0;                              // batch index inside the sub-tile
    const int cnt = __popc(bm_lo) + __popc(bm_hi);              // blended entries of this batch
    const bool flagged = (((lane < 32 ? bm_lo : bm_hi) >> (lane & 31)) & 1u) != 0u;
    const int below = mask_rank(bm_lo, bm_hi);
    const int dst = flagged ? below : cnt + (lane - below);     // a permutation of 0..63: blended entries first, in order
    const SubTile sub = decode_subtile(st, a.grid);
    const int pxi = sub.ox + (lane & 7), pyi = sub.oy + (lane >> 3);
    const bool inside = pxi < a.grid.W && pyi < a.grid.H;
    const float fx = (float)pxi, fy = (float)pyi;
    const float gr = p0.gr, gg = p0.gg, gb = p0.gb, gd = p0.gd, ga = p0.ga;

    stage_splat(s_b, dst, p0.r0, p0.r1, p0.r2);                 // unflagged lanes stage all-zero records (alpha 0) behind the list
    s_pslot[dst] = p0.pslot;
    s_pg[(lane >> 4) * 17 + (lane & 15)] = make_float4(gr, gg, gb, gd);

    float T = inside ? 1.0f : 0.0f, Tdead = 1.0f;               // blend.h: T = 0 for a pixel that has stopped
    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f;
    if (bq > 0) {
        T = fmaxf(p0.cs0, 0.0f);                                // (a stopped pixel is checkpointed as -T)
        sr = p0.cs1; sg = p0.cs2; sb = p0.cs3; sd = p0.cs4;
    }
    const float T_final = p0.cf0;
    // d/d(alpha_i) of [T_final * bg . g] and of [ga * (1 - T_final)]:  (T_final / (1 - alpha_i)) * (ga - bg.g)
    const float tail = T_final * (ga - (bg[0] * gr + bg[1] * gg + bg[2] * gb));
    float R = (p0.cf1 - sr) * gr + (p0.cf2 - sg) * gg + (p0.cf3 - sb) * gb - tail;
    if (HAS_DEPTH) R = fmaf(p0.cf4 - sd, gd, R);
    wave_lds_fence();
    // blend.h, conic_safe: the staged entries (compacted order) whose groups keep upstream's "power > 0" guard
    const unsigned long long unsafe = __builtin_amdgcn_ballot_w64(!conic_safe(s_b.ca[lane], s_b.cb[lane], s_b.cc[lane]));
    // staged entries (compacted order) that are trainable; a chunk without any skips phase B
    const unsigned long long need = PREFIX ? __ballot(lane < cnt && s_pslot[lane] != NO_SLOT) : ~0ull;

    const int g = XL::g(lane), h = XL::h(lane);                 // phase B role: splat g of the chunk, pixel group h
    float2* const xw_row = reinterpret_cast<float2*>(s_x + XL::wbase(lane));                            // phase A: my column
    const float4* const xr_row = reinterpret_cast<const float4*>(s_x + XL::rbase(lane));                // phase B: my row

    for (int c0 = 0; c0 < cnt; c0 += GC) {
        if (__all(T == 0.0f)) break;                            // (every pixel stopped inside this batch)
        const int cend = min(cnt, c0 + GC);
        // ---- phase A ---------------------------------------------------------------------------------
        {
            auto grad4 = [&](Alpha4& e, unsigned long long live_mask, const float4 (&col)[4], int k) {
                // e.Ag = A where this (live) pixel takes the splat: the product with dL/dalpha needs no `w > 0` compare and
                // select behind it.  (A pixel stopped BY one of the four: blend_group4 zeroes the gate with the weight.)
                float Tb[4], w[4];
                blend_group4(T, Tdead, live_mask, e.alpha, Tb, w, e.Ag);
                float2* x = xw_row + (k - c0) * (XL::WK / 2);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 c = col[u];
                    float cg = fmaf(c.z, gb, fmaf(c.y, gg, c.x * gr));
                    if (HAS_DEPTH) cg = fmaf(c.w, gd, cg);
                    R = fmaf(-cg, w[u], R);                                  // R_{i+1}
                    const float inv = __builtin_amdgcn_rcpf(1.0f - e.alpha[u]);
                    const float dLda = fmaf(Tb[u], cg, -(R * inv));
                    x[u * (XL::WK / 2)] = make_float2(e.Ag[u] * dLda, w[u]);
                }
            };
            auto group4 = [&](const Ops4& ops, int k) {
                float4 col[4] = {s_b.col[k], s_b.col[k + 1], s_b.col[k + 2], s_b.col[k + 3]};
                if (!HAS_DEPTH) { keep_b128(col[0]); keep_b128(col[1]); keep_b128(col[2]); keep_b128(col[3]); }
                const bool live = T > 0.0f;
                const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(live);
                Alpha4 e = splat_alpha4(ops, fx, fy, live);
                if ((unsafe >> k) & 0xfull) power_guard4(e, ops);
                grad4(e, live_mask, col, k);
            };
            // two groups per trip, ping-pong operand registers (next group's operands in flight, no register rotation)
            Ops4 opsA = load_ops4(s_b, c0);
            for (int k = c0; k < cend; k += 8) {
                const Ops4 opsB = load_ops4(s_b, (k + 4) & 63);
                group4(opsA, k);
                if (k + 4 >= cend) break;
                opsA = load_ops4(s_b, (k + 8) & 63);
                group4(opsB, k + 4);
            }
        }
        wave_lds_fence();
        // ---- phase B: lane = (splat g of the chunk, pixel group h = pixel rows 2h, 2h + 1) -------------------
        if (!PREFIX || ((need >> c0) & ((1ull << GC) - 1ull)) != 0ull) {
            const int kk = c0 + g;                              // rows >= cend hold stale data: never stored
            float S0 = 0.f, Sx = 0.f, Sxx = 0.f, Sy = 0.f, Sxy = 0.f, dr = 0.f, dg = 0.f, db = 0.f, dz = 0.f;
#pragma unroll
            for (int j = 0; j < GC / 2; ++j) {
                const float4 v = xr_row[j];                     // {aG, w} of pixels q = 2j, 2j + 1 of the group
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int q = 2 * j + t;
                    const float aG = t ? v.z : v.x, wq = t ? v.w : v.y;
                    const int pp = h * GC + q;                  // pixel of the sub-tile
                    float4 pg = s_pg[(pp >> 4) * 17 + (pp & 15)];
                    if (!HAS_DEPTH) keep_b128(pg);
                    const float X = (float)((q & 7) - 4);       // compile-time pixel coordinates (about column 4, row 2h)
                    S0 += aG;
                    if ((q & 7) != 4) { Sx = fmaf(aG, X, Sx); Sxx = fmaf(aG, X * X, Sxx); }
                    if (q >> 3) {
                        Sy += aG;
                        if ((q & 7) != 4) Sxy = fmaf(aG, X, Sxy);
                    }
                    dr = fmaf(wq, pg.x, dr); dg = fmaf(wq, pg.y, dg); db = fmaf(wq, pg.z, db);
                    if (HAS_DEPTH) dz = fmaf(wq, pg.w, dz);
                }
            }
            // moments about the splat centre: d = (gx, gy) - pixel;  with u = gx - (ox + 4), v = gy - (oy + 2h):
            //   sum aG dx = u S0 - Sx, sum aG dy = v S0 - Sy, sum aG dx^2 = u (u S0 - 2 Sx) + Sxx, ...  (Syy = Sy)
            const float u0 = s_b.px[kk] - (float)(sub.ox + 4), v0 = s_b.py[kk] - (float)(sub.oy + (GC / 8) * h);
            float mx = fmaf(u0, S0, -Sx), my = fmaf(v0, S0, -Sy);
            float mxx = fmaf(u0, mx - Sx, Sxx);
            float mxy = fmaf(v0, mx, fmaf(-u0, Sy, Sxy));
            float myy = fmaf(v0, my - Sy, Sy);
            float dop = S0;
            XL::reduce9(mx, my, mxx, mxy, myy, dop, dr, dg, db);
            if (HAS_DEPTH) dz = XL::reduce(dz);
            if (h == 0 && kk < cend) {
                // aG = (opacity G) dL/dalpha = dL/dG * G already (blend.h: A); dL/dopacity = sum of it / opacity
                const float inv_o = __builtin_amdgcn_exp2f(-s_b.op[kk]);
                const uint32_t ps = s_pslot[kk];
                if (!PREFIX || ps != NO_SLOT) {
                    char* dst = partial_at(prec, ps);
                    *reinterpret_cast<partial_v4*>(dst) = partial_v4{mx, my, mxx, mxy};
                    *reinterpret_cast<partial_v4*>(dst + 16) = partial_v4{myy, dop * inv_o, dr, dg};
                    if (PARTIAL_BYTES == 40) *reinterpret_cast<partial_v2*>(dst + 32) = partial_v2{db, dz};
                    else *reinterpret_cast<partial_v4*>(dst + 32) = partial_v4{db, dz, 0.f, 0.f};
                    if (PARTIAL_BYTES == 64) *reinterpret_cast<partial_v4*>(dst + 48) = partial_v4{0.f, 0.f, 0.f, 0.f};      // (the whole line: no partial sector)
                    touched[ps] = (uint8_t)1;
                }
            }
        }
        wave_lds_fence();                                       // the next chunk / batch overwrites the LDS buffers
    }
    // =============================================================================================================
    }   // active batch
    }   // slots of this wave
}

hipError_t launch_render_bwd(const RenderBwdArgs* a, int K, hipStream_t s) {
    uint64_t slots = 0;
    bool depth = false, prefix = false;
    for (int k = 0; k < K; ++k) {
        uint64_t mine = a[k].capacity / BATCH;                   // one wave per batch slot; the caller may know how many are in use
        if (a[k].used_slots && a[k].used_slots < mine) mine = a[k].used_slots;
        slots = mine > slots ? mine : slots;
        depth = depth || a[k].dL_ddepth != nullptr;
        prefix = prefix || a[k].grad_first > 0;
    }
    if (slots == 0) return hipSuccess;
    // Variants measured in rounds 2-5 and kept as BUILD-TIME macros (tools/build_variant.sh ... -DEXA_BWD_GC=16): chunk size 8
    // (default: 8 KB of LDS per wave = 5 waves per SIMD, DPP reductions) or 16 (13 KB, 3 waves per SIMD; C3: 52.5 vs 54.7 us);
    // EXA_BWD_SPW=2 batch slots per wave (with GC 16; 64.0 vs 55.6 us); EXA_BWD_WPB=2|4 independent waves per workgroup (slower);
    // EXA_BWD_LDS_PAD=<bytes> of unused dynamic LDS per workgroup (occupancy probe: 2.25 waves / SIMD = 60.6 us).
#ifndef EXA_BWD_GC
#define EXA_BWD_GC 8
#endif
#ifndef EXA_BWD_SPW
#define EXA_BWD_SPW 1
#endif
#ifndef EXA_BWD_WPB
#define EXA_BWD_WPB 1
#endif
#ifndef EXA_BWD_LDS_PAD
#define EXA_BWD_LDS_PAD 0
#endif
    constexpr int GC = EXA_BWD_GC, SPW = EXA_BWD_SPW, WPB = EXA_BWD_WPB;
    const Batch<RenderBwdArgs> b = make_batch(a, K);
    const dim3 grid((unsigned)((slots + SPW * WPB - 1) / (SPW * WPB)), K);
    if (prefix) {                       // (the prefix-aware instantiation takes one slot per wave)
        const dim3 g1((unsigned)((slots + WPB - 1) / WPB), K);
        if (depth) render_bwd_kernel<true, GC, 1, true, WPB><<<g1, RBLOCK * WPB, (size_t)EXA_BWD_LDS_PAD, s>>>(b);
        else render_bwd_kernel<false, GC, 1, true, WPB><<<g1, RBLOCK * WPB, (size_t)EXA_BWD_LDS_PAD, s>>>(b);
    } else if (depth) {
        render_bwd_kernel<true, GC, SPW, false, WPB><<<grid, RBLOCK * WPB, (size_t)EXA_BWD_LDS_PAD, s>>>(b);
    } else {
        render_bwd_kernel<false, GC, SPW, false, WPB><<<grid, RBLOCK * WPB, (size_t)EXA_BWD_LDS_PAD, s>>>(b);
    }
    return hipGetLastError();
}

}  // namespace exa
