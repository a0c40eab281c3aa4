// Per-sub-tile depth sort and front-to-back alpha compositing for gfx950.
//
// The blend runs ONE WAVE per 8x8-pixel sub-tile (lane l -> pixel (l & 7, l >> 3)), one wave per workgroup,
// no __syncthreads: a wave owns its list, its LDS slice and its 64 pixels, and frees its slot the moment
// it is done.  Split in two launches so each gets the shape it needs:
//   sort_subtiles_kernel  (256 threads, 16 KiB LDS): four waves co-operate on one sub-tile's bucket of
//       (depth bits << 32 | id) keys -- rank sorted runs of 64 merged by rank (binary search) in place in
//       LDS; lists > 2048 keys fall back to a register rank sort -- and write the sorted ids.
//       Third radix digit of the binning (binning.hip).  Its first 16 workgroups build the length-sorted
//       launch order of the blend.
//   render_fwd_kernel     (2.5 KiB LDS / wave): streams the sorted ids in batches of 64; each lane gathers ONE
//       64-byte splat record (48 B used) into the wave's LDS slice, SoA -- ids are fetched two batches ahead
//       and records one batch ahead, so the two dependent global round trips hide behind the blend of the
//       current batch -- then the 64 pixels blend the batch four splats at a time (blend.h).
//
// Replaces upstream SortPairs(depth digit) + renderCUDA (forward) of the rasterizer the reference
// calls at avatar/common/nets/module.py:632-640; per-pixel rule = oracle step 9/10
// (oracle/raster_oracle.py).
//
// Algorithmic HBM bytes: sort reads 8 B/instance, writes 4 B/instance; blend reads 4 B/instance + 48 B per
// gathered splat per sub-tile it touches (L2-resident after the first touch), writes 20 B/pixel
// (rgb, depth, alpha) + 20 B/pixel of checkpoint per batch entered (training only).
#include <stdlib.h>
#include "blend.h"

namespace exa {

constexpr int RBLOCK = 64;            // threads per workgroup of the per-pixel kernels: ONE wave

constexpr int SORT_TILE = 2048;       // keys of the LDS buffer (16 KiB)
constexpr int SBLOCK = 256;           // threads of a sort workgroup: four waves co-operate on ONE list
// (SPLIT_SORT_SUBTILES = 65536, common.h: from this many sub-tiles on the short lists get their own launch)

// ---- sort of lists up to SORT_TILE keys: rank-sorted runs of 64 + rank-based merges, in place in LDS -----
// A single wave issues roughly one VALU instruction per 5 cycles on gfx950 (probe: tools/probe/cmp_probe.hip),
// so the latency of a list is its instruction count; the longest lists (500-1000 keys) set the kernel time.
// Hence four waves share one list: thread t owns elements e * 256 + t.
//   Phase 1: every 64-key run is rank sorted by its own wave (each lane counts the keys of the run that are
//            smaller than its own; comparands are LDS broadcast reads, two keys per ds_read_b128).
//   Phase 2: runs are merged pairwise, doubling the run length per level; an element's merged position is
//            its offset in its own run plus its rank in the partner run (binary search; keys are unique:
//            the Gaussian id is the low word).  All searches of a level finish (barrier) before any write,
//            so the levels run IN PLACE.
template <int E>      // E = keys per thread: lists of up to 256 * E keys
__device__ __forceinline__ void lds_merge_sort(const unsigned long long* gkeys, int n, uint32_t* __restrict__ sorted,
                                               unsigned long long* keys_out, unsigned long long* buf, int tid) {
    constexpr int NPAD = SBLOCK * E;
    const int lane = tid & 63;
    unsigned long long key[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * SBLOCK + tid;
        key[e] = i < n ? gkeys[i] : ~0ull;                    // pad with +inf keys
        buf[i] = key[e];
    }
    __syncthreads();
    {   // phase 1: element i = e * 256 + tid lives in run (i >> 6); the E runs of a thread advance together
        uint32_t rank[E];
#pragma unroll
        for (int e = 0; e < E; ++e) rank[e] = 0;
        const ulonglong2* s2 = reinterpret_cast<const ulonglong2*>(buf);
#pragma unroll 4
        for (int j = 0; j < 32; ++j) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const ulonglong2 kk = s2[((e * SBLOCK + tid) >> 6) * 32 + j];
                rank[e] += (kk.x < key[e]) ? 1u : 0u;
                rank[e] += (kk.y < key[e]) ? 1u : 0u;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = e * SBLOCK + tid;
            // +inf pads compare equal to each other: give them distinct slots at the end of their run
            const uint32_t r = (i >= n) ? (uint32_t)lane : rank[e];
            buf[(i & ~63) + r] = key[e];
        }
    }
    __syncthreads();
    // phase 2: merge levels; the E binary searches of a thread run in lockstep (branch-free, fixed trip count)
#pragma unroll
    for (int L = 64, steps = 7; L < NPAD; L <<= 1, ++steps) {
        uint32_t lohi[E];                                      // lo | hi << 16
#pragma unroll
        for (int e = 0; e < E; ++e) {
            key[e] = buf[e * SBLOCK + tid];
            lohi[e] = (uint32_t)L << 16;
        }
        for (int it = 0; it < steps; ++it) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = e * SBLOCK + tid;
                const int base = i & ~(2 * L - 1);
                const bool first = (i - base) < L;
                const int p0 = base + (first ? L : 0);          // partner run
                const int lo = lohi[e] & 0xffff, hi = lohi[e] >> 16;
                const int mid = (lo + hi) >> 1;
                const bool act = lo < hi;
                const unsigned long long pk = buf[p0 + min(mid, L - 1)];
                // unique keys except the +inf pads: ties broken by run order so slots stay distinct
                const bool less = first ? (pk < key[e]) : (pk <= key[e]);
                const int nlo = (act && less) ? mid + 1 : lo;
                const int nhi = (act && !less) ? mid : hi;
                lohi[e] = (uint32_t)nlo | ((uint32_t)nhi << 16);
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = e * SBLOCK + tid;
            const int base = i & ~(2 * L - 1);
            const int off = i - base;
            buf[base + (off < L ? off : off - L) + (int)(lohi[e] & 0xffff)] = key[e];
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * SBLOCK + tid;
        if (i < n) {
            sorted[i] = (uint32_t)buf[i];
            if (keys_out) keys_out[i] = buf[i];               // (every input key was read before the first barrier)
        }
    }
}

// ---- bucket sort: the fast path for lists of 65 .. SORT_TILE keys -------------------------------------------
// The merge sort above spends ~600 thread-instructions per key (64 comparisons of the run rank sort + a dozen
// binary-search steps per merge level, all on 64-bit keys); the sort launch was VALU-bound on exactly that (9.3 M
// wave-instructions for 0.9 M keys).  The depths of one sub-tile's list are floats in a narrow range, so a
// DISTRIBUTION sort gets each key (almost) to its place in one pass:
//   1. block-wide min / max of the depth;  bucket = (depth - min) * (NB - 1) / (max - min), NB = padded list length
//      (monotone in the depth: a float subtraction, a multiplication by a positive scale and a truncation all are);
//   2. LDS histogram (one atomic per key, its return value is the key's arrival slot in the bucket), exclusive scan;
//   3. keys scattered into their bucket's range;
//   4. every key counts the keys of ITS bucket that are smaller (full 64-bit compare: depth bits, then Gaussian id) --
//      buckets hold ~1-5 keys -- and stores its id at bucket start + that count.
// The output is the exact total order by (depth bits, id) and does not depend on the atomics' arrival order.
// ~50 thread-instructions per key.  A list whose depths pile up in one bucket (> BUCKET_MAX keys: e.g. hundreds of splats at
// one depth) is left to the merge sort: returns false before anything was written.
constexpr int BUCKET_MAX = 48;
template <int E>      // E = keys per thread: lists of up to 256 * E keys, 256 * E buckets
__device__ __forceinline__ bool lds_bucket_sort(const unsigned long long* gkeys, int n, uint32_t* __restrict__ sorted,
                                                unsigned long long* keys_out, unsigned long long* buf, uint32_t* cnt,
                                                uint32_t* s_misc, int tid) {
    constexpr int NB = SBLOCK * E;
    const int lane = tid & 63, wave = tid >> 6;
    unsigned long long key[E];
    uint32_t dmin = 0xffffffffu, dmax = 0u;
    // all E loads first, unconditionally (clamped index), their first use in a loop of its own: with load and use in one
    // body the compiler waits for every key before it requests the next -- up to eight serial round trips (round 4 ISA)
#pragma unroll
    for (int e = 0; e < E; ++e) key[e] = gkeys[min(e * SBLOCK + tid, n - 1)];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * SBLOCK + tid;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[e] >> 32);
            dmin = min(dmin, d); dmax = max(dmax, d);
        } else {
            key[e] = ~0ull;
        }
        cnt[i] = 0u;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, d, 64));
        dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, d, 64));
    }
    if (lane == 0) { s_misc[wave] = dmin; s_misc[4 + wave] = dmax; }
    __syncthreads();
    dmin = min(min(s_misc[0], s_misc[1]), min(s_misc[2], s_misc[3]));
    dmax = max(max(s_misc[4], s_misc[5]), max(s_misc[6], s_misc[7]));
    // depths are positive floats: bit order == value order
    const float fmin = __uint_as_float(dmin);
    const float scale = (float)(NB - 1) / fmaxf(__uint_as_float(dmax) - fmin, 1e-30f);
    uint32_t bkt[E], slot[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        bkt[e] = 0u; slot[e] = 0u;
        if (e * SBLOCK + tid < n) {
            const float f = (__uint_as_float((uint32_t)(key[e] >> 32)) - fmin) * scale;
            bkt[e] = (uint32_t)min(NB - 1, (int)f);
            slot[e] = __hip_atomic_fetch_add(&cnt[bkt[e]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    // exclusive scan of the NB counters (thread t owns counters t * E .. t * E + E - 1) and the largest bucket
    uint32_t c[E], sum = 0u, mx = 0u;
#pragma unroll
    for (int e = 0; e < E; ++e) { c[e] = cnt[tid * E + e]; sum += c[e]; mx = max(mx, c[e]); }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += o;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if (lane == 63) s_misc[8 + wave] = incl;
    if (lane == 0) s_misc[12 + wave] = mx;
    __syncthreads();
    mx = max(max(s_misc[12], s_misc[13]), max(s_misc[14], s_misc[15]));
    if (mx > (uint32_t)BUCKET_MAX) return false;                 // workgroup-uniform; nothing written yet
    uint32_t run = incl - sum;
    for (int w2 = 0; w2 < wave; ++w2) run += s_misc[8 + w2];
#pragma unroll
    for (int e = 0; e < E; ++e) { cnt[tid * E + e] = run; run += c[e]; }      // counters -> bucket starts
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (e * SBLOCK + tid < n) buf[cnt[bkt[e]] + slot[e]] = key[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (e * SBLOCK + tid < n) {
            const uint32_t s0 = cnt[bkt[e]], s1 = bkt[e] + 1u < (uint32_t)NB ? cnt[bkt[e] + 1u] : (uint32_t)n;
            uint32_t rank = s0;
            for (uint32_t j = s0; j < s1; ++j) rank += buf[j] < key[e] ? 1u : 0u;
            sorted[rank] = (uint32_t)key[e];
            if (keys_out) keys_out[rank] = key[e];            // in place over the input: all keys sit in registers by now
        }
    }
    return true;
}

// Single-wave variant for short lists (<= 64 keys): one rank-sort pass, no barriers.
__device__ __forceinline__ void wave_rank_sort64(const unsigned long long* gkeys, int n, uint32_t* __restrict__ sorted,
                                                 unsigned long long* keys_out, unsigned long long* buf, int lane) {
    const unsigned long long mine = lane < n ? gkeys[lane] : ~0ull;
    buf[lane] = mine;
    wave_lds_fence();
    const ulonglong2* s2 = reinterpret_cast<const ulonglong2*>(buf);
    uint32_t rank = 0;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
        const ulonglong2 kk = s2[j];
        rank += (kk.x < mine) ? 1u : 0u;
        rank += (kk.y < mine) ? 1u : 0u;
    }
    if (lane < n) {
        sorted[rank] = (uint32_t)mine;
        if (keys_out) keys_out[rank] = mine;                  // (the wave's loads are complete: rank depends on all of them)
    }
}

// ---- lists longer than SORT_TILE: rank sort, R own keys per thread in registers per pass, comparands staged
// in LDS 1024 at a time.  O(n^2 / 256) but any length and no global scratch; such lists are rare.
template <int R>
__device__ __forceinline__ void rank_sort_list(const unsigned long long* __restrict__ gkeys, int n, int first,
                                               uint32_t* __restrict__ sorted, unsigned long long* s_keys, int tid) {
    unsigned long long mine[R];
    uint32_t rank[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = first + r * SBLOCK + tid;
        mine[r] = i < n ? gkeys[i] : ~0ull;
        rank[r] = 0;
    }
    for (int tile = 0; tile < n; tile += SORT_TILE) {
        const int tn = min(SORT_TILE, n - tile);
        __syncthreads();
        for (int i = tid; i < ((tn + 1) & ~1); i += SBLOCK) s_keys[i] = i < tn ? gkeys[tile + i] : ~0ull;
        __syncthreads();
        const ulonglong2* s2 = reinterpret_cast<const ulonglong2*>(s_keys);
#pragma unroll 4
        for (int j = 0; j < (tn + 1) / 2; ++j) {
            const ulonglong2 kk = s2[j];                      // uniform address: LDS broadcast read
#pragma unroll
            for (int r = 0; r < R; ++r) {
                rank[r] += (kk.x < mine[r]) ? 1u : 0u;
                rank[r] += (kk.y < mine[r]) ? 1u : 0u;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (first + r * SBLOCK + tid < n) sorted[rank[r]] = (uint32_t)mine[r];
}

// ---- launch order of the forward blend (see length_class / xcd_region in common.h) -------------------------
// Runs as the first ORDER_WGS workgroups of the sort launch, i.e. concurrently with the sorting and off the
// critical path.  Every ordering workgroup histograms the (XCD region, list-length class) bins of the sub-tiles of all
// active cells, then ranks its own share inside LDS, reserves one contiguous range per bin with a single device atomic, and
// writes the records.
//
// XCD-AWARE ORDER.  Workgroup b of a launch runs on XCD b mod 8 and every XCD has its own L2.  A splat record is shared by the
// ~4 neighbouring sub-tiles it reaches; one length-sorted sequence deals those neighbours to different XCDs, so the record
// comes in from beyond L2 once per sub-tile (and the backward's 40-byte partial records and `touched` bytes, which neighbours
// write into the same 32-byte sectors, are written back once per XCD).  The order is therefore EIGHT interleaved
// length-sorted streams: sub-tile st belongs to region xcd_region(st) (blocks of 2 x 4 sub-tiles: every 64 x 64 cell gives
// exactly eight sub-tiles to every region), and the k-th sub-tile of region x in descending list length sits at
// position 8 k + x.  Every region owns exactly subtiles / 8 positions, so the records stay a permutation; heavy-first holds
// inside every XCD, which is where it matters (an XCD's waves are dispatched in launch order).
constexpr int ORDER_WGS = 32;
constexpr int ORDER_BINS = XCD_REGIONS * ORDER_CLASSES;
constexpr int ORDER_LDS_WORDS = 3 * ORDER_BINS + XCD_REGIONS * (BWD_ORDER_DEPTH + 1);
// The same workgroups write the launch order of the BACKWARD blend (one wave per 64-entry batch, render_bwd.hip):
// batch-major -- the first batches of all lists, longest list first, then all second batches, ... -- which is heavy first
// without knowing the blended counts: the front batches of a list are the ones whose entries get blended, the deep ones
// the forward often does not even enter.  In slot order the launch ended with whatever the last cells held (a full batch
// takes 18 us at five waves per SIMD, started as late as 35 us in) and spent a third of its dispatches on slots without
// work (end slots, padding: 14 k of 24 k).  Per region x: position of batch b of the list at descending rank k = 8 (B_b + k) + x,
// with B_b = sum over b' < b of the number of the region's lists with more than b' batches -- all from the histogram, no
// sort.  The regions' streams differ in length by a few per cent: row kb of the interleaved streams holds only the streams
// longer than kb, i.e. batch kb of region x sits at sum over x' of min(M_x', kb) + (number of x' < x with M_x' > kb) -- 8 kb + x
// while every stream is alive, a bijection onto [0, sum M) always (the rows behind the shortest stream's end lose the
// alignment with the XCDs, nothing else).  Batches from BWD_ORDER_DEPTH on (lists of > 1024 entries) are appended behind
// through a counter.
template <typename RangeOf>       // RangeOf(st) -> [begin, end) of sub-tile st's list; the records hold what it returns
__device__ __forceinline__ void order_slots(const TileWs& w, uint32_t* __restrict__ bwd_order, int subtiles, int part, int tid,
                                            uint32_t* __restrict__ lds, RangeOf range_of) {
    // (LDS lent by the caller: the sort's own buffers -- arrays of its own would cost the sort kernel its sixth wave per SIMD)
    uint32_t* const s_off = lds;                                  // [region][class] sub-tiles of the region in longer classes
    uint32_t* const s_cnt = lds + ORDER_BINS;
    uint32_t* const s_base = lds + 2 * ORDER_BINS;
    uint32_t* const s_bbase = lds + 3 * ORDER_BINS;               // [region][BWD_ORDER_DEPTH + 1]
    const int lane = tid & 63;
#ifdef EXA_PROBE_SORTLINE  // probe build only (tools/gpu_sort_timeline.py): the phases of every ordering workgroup, 100 MHz clock
#define ORDER_STAMP(ph) do { if (tid == 0) w.part_cnt[2 * (subtiles + ORDER_WGS) + part * 8 + (ph)] = (uint32_t)wall_clock64(); } while (0)
#else
#define ORDER_STAMP(ph) do {} while (0)
#endif
    ORDER_STAMP(0);
    // Histogram over ALL sub-tiles from the class codes the binning left next to the ranges: sixteen lists per 16-byte load,
    // one round trip per 16 384 sub-tiles.  (Until round 6: the ranges of the active cells, found through cell_desc -- two
    // dependent loads per trip, three trips for an avatar view, 5.9 of the 9.5 us these workgroups ran.)
    for (int i = tid; i < ORDER_BINS; i += SBLOCK) s_cnt[i] = 0u;
    __syncthreads();
    const uint4* __restrict__ codes = reinterpret_cast<const uint4*>(w.cls_code);
    const int n16 = subtiles / 16;
    for (int base = 0; base < n16; base += SBLOCK * 4) {
        uint4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = base + i * SBLOCK + tid;
            v[i] = idx < n16 ? codes[idx] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = base + i * SBLOCK + tid;                 // sub-tiles 16 idx .. 16 idx + 15: a quarter of a cell
            const uint32_t word[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            if ((word[0] | word[1] | word[2] | word[3]) == 0u) continue;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t cls = (word[j >> 2] >> (8 * (j & 3))) & 0xffu;
                if (cls) atomicAdd(&s_cnt[xcd_region((idx & 3) * 16 + j) * ORDER_CLASSES + cls], 1u);
            }
        }
    }
    __syncthreads();
    ORDER_STAMP(1);
    // per region: exclusive prefix of the histogram, longest class first, class 0 (empty) last; a wave takes two regions
    // (side by side: the twelve shuffles of one region are a dependent chain)
    {
        constexpr int RPW = (XCD_REGIONS + SBLOCK / 64 - 1) / (SBLOCK / 64);       // regions per wave
        const int cls = lane == 63 ? 0 : 63 - lane;
        uint32_t v[RPW], incl[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int x = (tid >> 6) + j * (SBLOCK / 64);
            v[j] = incl[j] = x < XCD_REGIONS ? s_cnt[x * ORDER_CLASSES + cls] : 0u;
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const uint32_t o = __shfl_up(incl[j], d, 64);
                if (lane >= d) incl[j] += o;
            }
        }
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int x = (tid >> 6) + j * (SBLOCK / 64);
            if (x < XCD_REGIONS) s_off[x * ORDER_CLASSES + cls] = incl[j] - v[j];
        }
    }
    __syncthreads();
    for (int i = tid; i < ORDER_BINS; i += SBLOCK) s_cnt[i] = 0u;
    if (tid < XCD_REGIONS) {   // lists of the region with more than b batches = lists of a class above 4 b = s_off[4 b]
        uint32_t more[BWD_ORDER_DEPTH];                           // (all loads first: a store in between would order them)
#pragma unroll
        for (int b = 0; b < BWD_ORDER_DEPTH; ++b) more[b] = s_off[tid * ORDER_CLASSES + 4 * b];
        uint32_t run = 0u;
#pragma unroll
        for (int b = 0; b < BWD_ORDER_DEPTH; ++b) { s_bbase[tid * (BWD_ORDER_DEPTH + 1) + b] = run; run += more[b]; }
        s_bbase[tid * (BWD_ORDER_DEPTH + 1) + BWD_ORDER_DEPTH] = run;
    }
    __syncthreads();
    uint32_t M[XCD_REGIONS];                                      // batches in every region's stream
#pragma unroll
    for (int x = 0; x < XCD_REGIONS; ++x) M[x] = s_bbase[x * (BWD_ORDER_DEPTH + 1) + BWD_ORDER_DEPTH];
    uint32_t m_min = M[0], deep_base = M[0];                      // deep_base: first position behind the eight streams
#pragma unroll
    for (int x = 1; x < XCD_REGIONS; ++x) { m_min = min(m_min, M[x]); deep_base += M[x]; }
    if (part == 0 && tid == 0) { w.bwd_meta[0] = deep_base; w.bwd_meta[2] = bwd_order ? BWD_ORDER_MAGIC : 0u; }
    ORDER_STAMP(2);
    const int per = (subtiles + ORDER_WGS - 1) / ORDER_WGS;
    const int lo = part * per, hi = min(subtiles, lo + per);
    for (int base = lo; base < hi; base += SBLOCK * 4) {
        uint2 r[4];
        uint32_t rank[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int st = base + i * SBLOCK + tid;
            r[i] = st < hi ? range_of(st) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int st = base + i * SBLOCK + tid;
            const bool valid = st < hi;
            const int cls = length_class(r[i].y - r[i].x), x = xcd_region(st & 63);
            // the (many) empty sub-tiles: one LDS atomic per wave and region, all regions' leaders in the same instruction
            const bool is_empty = valid && cls == 0;
            unsigned long long mine = 0ull;                     // the empty lanes of this lane's region
#pragma unroll
            for (int xr = 0; xr < XCD_REGIONS; ++xr) {
                const unsigned long long m = __ballot(is_empty && x == xr);
                if (x == xr) mine = m;
            }
            const int leader = is_empty ? __ffsll((long long)mine) - 1 : lane;
            uint32_t b0 = 0u;
            if (cls) b0 = atomicAdd(&s_cnt[x * ORDER_CLASSES + cls], 1u);
            else if (is_empty && lane == leader) b0 = atomicAdd(&s_cnt[x * ORDER_CLASSES], (uint32_t)__popcll(mine));
            b0 = (uint32_t)__shfl((int)b0, leader, 64);
            rank[i] = b0 + (is_empty ? (uint32_t)__popcll(mine & ((1ull << lane) - 1ull)) : 0u);
        }
        ORDER_STAMP(3);
        __syncthreads();
        {   // one device atomic per bin this workgroup holds anything of (all of a thread's requests in flight together)
            constexpr int BPT = (ORDER_BINS + SBLOCK - 1) / SBLOCK;
            uint32_t c[BPT], got[BPT];
#pragma unroll
            for (int j = 0; j < BPT; ++j) c[j] = j * SBLOCK + tid < ORDER_BINS ? s_cnt[j * SBLOCK + tid] : 0u;
#pragma unroll
            for (int j = 0; j < BPT; ++j) got[j] = c[j] ? atomicAdd(&w.cls_cur[j * SBLOCK + tid], c[j]) : 0u;
#pragma unroll
            for (int j = 0; j < BPT; ++j)
                if (j * SBLOCK + tid < ORDER_BINS) s_base[j * SBLOCK + tid] = s_off[j * SBLOCK + tid] + got[j];
        }
        __syncthreads();
        ORDER_STAMP(4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int st = base + i * SBLOCK + tid;
            if (st < hi) {
                const int cls = length_class(r[i].y - r[i].x), x = xcd_region(st & 63);
                const uint32_t k = s_base[x * ORDER_CLASSES + cls] + rank[i];        // rank inside the region's stream
                w.slots[k * XCD_REGIONS + x] = make_uint4(r[i].x, r[i].y, (uint32_t)st, 0u);
                const uint32_t nb = (r[i].y - r[i].x + BATCH - 1) / BATCH, slot0 = r[i].x / BATCH;
                for (uint32_t b = 0; bwd_order && b < nb; ++b) {
                    uint32_t q;
                    if (b < (uint32_t)BWD_ORDER_DEPTH) {          // row kb of the interleaved streams, compacted over the ended ones
                        const uint32_t kb = s_bbase[x * (BWD_ORDER_DEPTH + 1) + b] + k;
                        q = kb * XCD_REGIONS + (uint32_t)x;       // every stream alive: the row is complete
                        if (kb >= m_min) {
                            q = 0u;
#pragma unroll
                            for (int xo = 0; xo < XCD_REGIONS; ++xo) q += min(M[xo], kb) + (xo < x && M[xo] > kb ? 1u : 0u);
                        }
                    } else {
                        q = deep_base + atomicAdd(&w.bwd_meta[1], 1u);
                    }
                    bwd_order[q] = slot0 + b;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < ORDER_BINS; i += SBLOCK) s_cnt[i] = 0u;
        __syncthreads();
        ORDER_STAMP(5);
    }
#undef ORDER_STAMP
}

// Short lists of a LARGE image (>= SPLIT_SORT_SUBTILES sub-tiles, e.g. 2048 x 2048 px): one wave per sub-tile, four
// independent sub-tiles per workgroup, 512 B of LDS per wave.  With content everywhere (C5: a background scene behind the
// avatar) most of the 65 536 lists hold a few dozen keys, and one 256-thread workgroup with 24 KiB of LDS per list -- six
// per CU, three of the four waves idle -- spent 160 us on them; sort_subtiles_kernel<true> then only takes the lists
// longer than one wave and skips cells without any after ONE scalar load.
__global__ __launch_bounds__(SBLOCK) void sort_short_kernel(Batch<RenderFwdArgs> batch) {
    __shared__ __attribute__((aligned(16))) unsigned long long s_buf[(SBLOCK / 64) * 64];
    const RenderFwdArgs& a = batch.v[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int idx = (int)blockIdx.x * (SBLOCK / 64) + wave;     // sub-tile in cell_desc order
    if (idx >= a.grid.subtiles) return;
    if (idx >= (int)a.tw.header->active_cells * SUBS_PER_CELL) return;
    const int st = (int)a.tw.cell_desc[idx >> 6].x * SUBS_PER_CELL + (idx & 63);
    const uint2 range = a.tw.ranges[st];
    const int n = (int)(range.y - range.x);
    if (n > 0 && n <= 64)
        wave_rank_sort64(a.bw.keys + range.x, n, a.bw.sorted + range.x, a.keep_sorted_keys ? a.bw.keys + range.x : nullptr,
                         s_buf + wave * 64, lane);
}

// KEEP = true: merge source of a composite render (compose.hip): the sorted 64-bit keys replace the unsorted ones, in place.
// Its own instantiation: as a run-time flag the extra stores cost the headline sort 1.4 us of 15.1 on C3.
// Six waves per SIMD (<= 85 VGPRs; the eight-keys-per-thread paths of the long lists spill ~50 bytes): the launch is a
// stream of ~3 700 three-microsecond workgroups for an avatar view, and at the 104 registers the compiler takes on its own
// only four of them fit a CU -- the timeline of the launch (tools/gpu_sort_timeline.py) showed workgroups with a list still
// STARTING 11.7 us in.  Six is also what the 25 KiB of LDS allow.  16.0 -> 13.6 us by events (five waves: 14.4), C3 +1.8 %.
template <bool SPLIT, bool KEEP>
__global__ __launch_bounds__(SBLOCK) __attribute__((amdgpu_waves_per_eu(6, 6))) void sort_subtiles_kernel(Batch<RenderFwdArgs> batch) {
    __shared__ __attribute__((aligned(16))) unsigned long long s_buf[SORT_TILE];
    __shared__ uint32_t s_cnt[SORT_TILE];                        // bucket counters / starts of the distribution sort
    __shared__ uint32_t s_misc[16];
    const RenderFwdArgs& a = batch.v[blockIdx.y];
    if ((int)blockIdx.x >= a.grid.subtiles + ORDER_WGS) return;   // a job with a smaller image than the largest of the batch
    const int tid = threadIdx.x;
#ifdef EXA_PROBE_SORTLINE   // probe build only (tools/gpu_sort_timeline.py): start / end of every workgroup, 100 MHz clock
    struct TL { const RenderFwdArgs& a; unsigned long long t0; int tid;
        __device__ ~TL() { __syncthreads(); if (tid == 0) { a.tw.part_cnt[2 * blockIdx.x] = (uint32_t)t0; a.tw.part_cnt[2 * blockIdx.x + 1] = (uint32_t)wall_clock64(); } } } tl{a, (unsigned long long)wall_clock64(), tid};
#endif
    if (blockIdx.x < ORDER_WGS) {
        // (the backward's order only for renders that keep their context: a no_grad frame has no backward)
        static_assert(ORDER_LDS_WORDS <= SORT_TILE, "order_slots borrows the bucket counters of the sort");
        order_slots(a.tw, a.store_ctx ? reinterpret_cast<uint32_t*>(a.bw.bucket) : nullptr, a.grid.subtiles, (int)blockIdx.x, tid,
                    s_cnt, [&](int st) { return a.tw.ranges[st]; });
        return;
    }
#ifdef EXA_PROBE_SORT
    const unsigned long long t0 = __builtin_readcyclecounter();
#endif
    if (blockIdx.x == ORDER_WGS && tid == 0 && (uint64_t)a.tw.header->num_rendered > a.capacity) a.tw.header->overflow = 1u;
    // Only the sub-tiles of the ACTIVE cells hold lists (cell_desc lists those cells first, heaviest first;
    // header.active_cells counts them): an avatar view has ~3 800 non-empty lists in 16 384 sub-tiles, and the
    // workgroups of the empty ones leave after ONE scalar load instead of two dependent vector loads.  (A
    // grid-stride loop over the active sub-tiles with a smaller grid cost 2.5x the registers and ran slower; so did, in
    // round 3, a grid over 3/8 of the cells whose workgroups continue with further lists only when more cells are
    // active: the loop around the sorts spills on the hot path at six waves per SIMD -- 13.4 -> 48.9 us -- although the
    // timeline shows the dispatch of the 12 600 idle workgroups lasting as long as the sorting itself.)
    const int wg = (int)blockIdx.x - ORDER_WGS;
    if (wg >= (int)a.tw.header->active_cells * SUBS_PER_CELL) return;
    if (SPLIT && a.tw.cell_long[wg >> 6] == 0u) return;         // the same word for the 64 workgroups of a cell
    const int st = (int)a.tw.cell_desc[wg >> 6].x * SUBS_PER_CELL + (wg & 63);
    const uint2 range = a.tw.ranges[st];
    const int n = (int)(range.y - range.x);                     // workgroup-uniform; empty on overflow
    if (n == 0) return;
    unsigned long long* gkeys = a.bw.keys + range.x;
    uint32_t* sorted = a.bw.sorted + range.x;
    unsigned long long* const keys_out = KEEP ? gkeys : nullptr;
    if (n <= 64) {
        if (!SPLIT && tid < 64) wave_rank_sort64(gkeys, n, sorted, keys_out, s_buf, tid);      // (SPLIT: sort_short_kernel did it)
        return;
    }
    if (n > SORT_TILE) {   // longer lists: 2048 own keys at a time against the whole list (any length)
        for (int first = 0; first < n; first += 8 * SBLOCK) rank_sort_list<8>(gkeys, n, first, sorted, s_buf, tid);
        if (keys_out) {    // the passes re-read the unsorted keys, so the sorted ones are rebuilt afterwards: depth bits of the record
            __threadfence_block();
            __syncthreads();
            for (int i = tid; i < n; i += SBLOCK) {
                const uint32_t id = sorted[i];
                keys_out[i] = ((unsigned long long)__float_as_uint(a.splats[id].depth) << 32) | id;
            }
        }
        return;
    }
    // distribution sort first; the merge sort takes the (rare) lists whose depths pile up in one bucket
    bool done;
    if (n <= 256) done = lds_bucket_sort<1>(gkeys, n, sorted, keys_out, s_buf, s_cnt, s_misc, tid);
    else if (n <= 512) done = lds_bucket_sort<2>(gkeys, n, sorted, keys_out, s_buf, s_cnt, s_misc, tid);
    else if (n <= 1024) done = lds_bucket_sort<4>(gkeys, n, sorted, keys_out, s_buf, s_cnt, s_misc, tid);
    else done = lds_bucket_sort<8>(gkeys, n, sorted, keys_out, s_buf, s_cnt, s_misc, tid);
    if (done) return;
    __syncthreads();
    if (n <= 256) lds_merge_sort<1>(gkeys, n, sorted, keys_out, s_buf, tid);             // (a plain O(n^2) rank sort of the whole
    else if (n <= 512) lds_merge_sort<2>(gkeys, n, sorted, keys_out, s_buf, tid);        //  list was 40 % slower: LDS-pipe bound)
    else if (n <= 1024) lds_merge_sort<4>(gkeys, n, sorted, keys_out, s_buf, tid);
    else lds_merge_sort<8>(gkeys, n, sorted, keys_out, s_buf, tid);
#ifdef EXA_PROBE_SORT
    __syncthreads();
    if (tid == 0) a.tw.part_cnt[st] = (uint32_t)(__builtin_readcyclecounter() - t0);      // probe build only
#endif
}

// ---- blend ---------------------------------------------------------------------------------------------
// TWO = true: composite render (compose.hip).  The sorted list holds ids of TWO finished renders of this camera (bit 31 =
// source B) and the launch records carry the sub-tile only (its range is read from tw.ranges); everything else -- batches,
// checkpoints, blended masks -- is the plain kernel, its own instantiation so that the headline path carries no test.
template <bool STORE, bool TWO>
__global__ __launch_bounds__(RBLOCK) void render_fwd_kernel(Batch<RenderFwdArgs> batch) {
    __shared__ BatchLds s_b;

    const RenderFwdArgs& a = batch.v[blockIdx.y];
    if ((int)blockIdx.x >= a.grid.subtiles) return;
    const int lane = threadIdx.x;
#ifdef EXA_PROBE_FWD       // probe build only (tools/gpu_fwd_timeline.py): start / end of every wave on the chip-wide 100 MHz clock
    const unsigned long long t0 = wall_clock64();
#endif
    const uint4 slot = a.tw.slots[blockIdx.x];                  // {begin, end, st, 0}; empty range on overflow
    const SubTile sub = decode_subtile((int)slot.z, a.grid);
    const int st = sub.st;
    if (sub.ox >= a.grid.W || sub.oy >= a.grid.H) return;       // padding sub-tile of a border cell
    const int pxi = sub.ox + (lane & 7), pyi = sub.oy + (lane >> 3);
    const bool inside = pxi < a.grid.W && pyi < a.grid.H;
    const float fx = (float)pxi, fy = (float)pyi;
    const uint2 range = TWO ? a.tw.ranges[slot.z] : make_uint2(slot.x, slot.y);
    const int n = (int)(range.y - range.x);
    if (TWO && n == 0 && reuse_a_pixels(a.src_color, a.src_bg, a.bg)) {
        // no entry of B in this sub-tile (compose.hip emptied its list): the composite's pixels are source A's own, which
        // blended exactly the list the merge would have produced, over an equal background
        if (inside) {
            const size_t HW = (size_t)a.grid.W * a.grid.H, pix = (size_t)pyi * a.grid.W + pxi;
            const float c0 = a.src_color[pix], c1 = a.src_color[HW + pix], c2 = a.src_color[2 * HW + pix];
            const float d = a.src_depth[pix], al = a.src_alpha[pix];
            a.out_color[pix] = c0; a.out_color[HW + pix] = c1; a.out_color[2 * HW + pix] = c2;
            a.out_depth[pix] = d; a.out_alpha[pix] = al;
        }
        if (STORE && lane == 0) a.tw.fwd_exit[st] = make_uint2(0u, 0u);
        return;
    }

    float T = inside ? 1.0f : 0.0f, Tdead = 1.0f;              // blend.h: T = 0 once the pixel has stopped, Tdead = what it stopped with
    v2f Crg = {0.f, 0.f}, Cbd = {0.f, 0.f};                     // (r, g) and (b, depth) accumulators
    const Splat* __restrict__ splats = a.splats;
    const uint32_t* __restrict__ sorted = a.bw.sorted + range.x;
    auto record = [&](uint32_t id) -> const float4* {
        if (TWO) return reinterpret_cast<const float4*>(((id & SRC_B) ? a.splats2 : splats) + (id & ~SRC_B));
        return reinterpret_cast<const float4*>(splats + id);
    };

    // software pipeline: ids two batches ahead, records one batch ahead
    uint32_t id_next = 0;
    float2 r0 = make_float2(0.f, 0.f);
    float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f), r2 = r1;
    if (n > 0) {
        const uint32_t id0 = lane < n ? sorted[lane] : 0u;
        if (64 + lane < n) id_next = sorted[64 + lane];
        if (lane < n) {
            const float4* rec = record(id0);
            r0 = *reinterpret_cast<const float2*>(rec); r1 = rec[1]; r2 = rec[2];
        }
    }
    // training: per-pixel state at the START of every batch slot (and at the exit), so that the backward
    // pass can give every batch its own wave (render_bwd.hip).  A stopped pixel is stored as -T.
    float* ckpt = a.bw.ckpt + (size_t)(range.x / BATCH) * (5 * 64) + lane;
    int entered = 0;
    for (int base = 0; base < n; base += 64) {
        if (__all(T == 0.0f)) break;
        if (STORE && base > 0) {
            float* c = ckpt + (size_t)entered * (5 * 64);
            c[0] = T > 0.0f ? T : -Tdead; c[64] = Crg.x; c[128] = Crg.y; c[192] = Cbd.x; c[256] = Cbd.y;
        }
        ++entered;
        // blend.h, conic_safe: the entries whose groups keep upstream's "power > 0" guard (none, in any sane scene)
        const unsigned long long unsafe = __builtin_amdgcn_ballot_w64(!conic_safe(r1.x, r1.y, r1.z));
        stage_splat(s_b, lane, r0, r1, r2);
        // issue the next batch's gathers and the ids of the batch after it; lanes past the end of the list
        // stage an all-zero record (opacity 0 -> alpha 0), so the blend below always runs whole groups of four
        {
            const int jn = base + 64 + lane;
            r0 = make_float2(0.f, 0.f); r1 = make_float4(0.f, 0.f, 0.f, 0.f); r2 = r1;
            if (jn < n) {
                const float4* rec = record(id_next);
                r0 = *reinterpret_cast<const float2*>(rec); r1 = rec[1]; r2 = rec[2];
            }
            if (jn + 64 < n) id_next = sorted[jn + 64];
        }
        wave_lds_fence();
        const int cnt = min(64, n - base);
        // training: which entries of this batch were blended into at least one pixel (alpha > 0 at a live pixel; the
        // entry that stops a pixel counts).  The backward pass replays exactly those: on avatar-like scenes only
        // ~60 % of the entries of the batches the forward enters (tools/cpu_blend_stats.py).
        unsigned long long blended = 0ull;
        // Blend one group of four splats whose alphas are known; returns true when every pixel of the sub-tile has stopped.
        auto blend4 = [&](const Alpha4& e, const float4& c0, const float4& c1, const float4& c2, const float4& c3, int k) -> bool {
            // (no wave-wide vote on "does any live pixel take any of the four": with exact-footprint lists nearly every
            //  group is taken by some pixel, and the vote cost more than the blends it skipped: 42.8 -> 41.8 us on C3)
#ifdef EXA_FWD_SKIP_VOTE
            const float amax = fmaxf(fmaxf(e.alpha[0], e.alpha[1]), fmaxf(e.alpha[2], e.alpha[3])) * T;
            if (!__any(amax > 0.0f)) return false;
#endif
            const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(T > 0.0f);      // before the group
            if (STORE) {
                // (BEFORE the blend: in the basic block of the compares -- behind the blend's rare branch the masks come back
                //  through VGPRs.)  Four votes -> one nibble, all of it on the scalar unit: the blend needs the compares behind "not skipped" and
                // "was live" anyway, a ballot of a compare IS its SGPR mask, and the rest is s_and / s_cmp / s_cselect.  (HIP's
                // __ballot compares an int against 0, i.e. takes the mask through a VGPR; left to the compiler the nibble was
                // assembled in a VGPR: a v_cndmask, three v_or and a v_readfirstlane per group of four.)
                auto vote_bit = [&](int j, uint32_t bit) -> uint32_t {
                    const unsigned long long m = e.took[j] & live_mask;
                    uint32_t r;
                    asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, %2, 0" : "=s"(r) : "s"(m), "s"(bit) : "scc");
                    return r;
                };
                const uint32_t nib = vote_bit(0, 1u) | vote_bit(1, 2u) | vote_bit(2, 4u) | vote_bit(3, 8u);
                blended |= (unsigned long long)nib << k;
            }
            float Tb[4], w[4];
            blend_group4(T, Tdead, live_mask, e.alpha, Tb, w);
            Crg = __builtin_elementwise_fma(v2f{c0.x, c0.y}, v2f{w[0], w[0]}, Crg);
            Cbd = __builtin_elementwise_fma(v2f{c0.z, c0.w}, v2f{w[0], w[0]}, Cbd);
            Crg = __builtin_elementwise_fma(v2f{c1.x, c1.y}, v2f{w[1], w[1]}, Crg);
            Cbd = __builtin_elementwise_fma(v2f{c1.z, c1.w}, v2f{w[1], w[1]}, Cbd);
            Crg = __builtin_elementwise_fma(v2f{c2.x, c2.y}, v2f{w[2], w[2]}, Crg);
            Cbd = __builtin_elementwise_fma(v2f{c2.z, c2.w}, v2f{w[2], w[2]}, Cbd);
            Crg = __builtin_elementwise_fma(v2f{c3.x, c3.y}, v2f{w[3], w[3]}, Crg);
            Cbd = __builtin_elementwise_fma(v2f{c3.z, c3.w}, v2f{w[3], w[3]}, Cbd);
            // "every pixel of the sub-tile dead" is looked for once per EIGHT splats (a dead pixel takes alpha * 0 = 0,
            // so walking four more splats changes nothing): one vote less per group, 42.3 vs 42.8 us on C3
#ifdef EXA_FWD_EXIT4
            return __all(T == 0.0f);
#else
            return (k & 4) ? __all(T == 0.0f) : false;
#endif
        };
        // two groups per trip with ping-pong operand registers: the operands of the next group are in flight during the
        // current one, and no register-to-register rotation is needed (a `cur = nxt` copy cost 12 v_mov_b64 per group)
        auto group4 = [&](const Ops4& ops, int k) -> bool {
            const float4 c0 = s_b.col[k], c1 = s_b.col[k + 1], c2 = s_b.col[k + 2], c3 = s_b.col[k + 3];
            Alpha4 e = splat_alpha4(ops, fx, fy, true);
            if ((unsafe >> k) & 0xfull) power_guard4(e, ops);
            return blend4(e, c0, c1, c2, c3, k);
        };
        Ops4 opsA = load_ops4(s_b, 0);
        for (int k = 0; k < cnt; k += 8) {
            const Ops4 opsB = load_ops4(s_b, (k + 4) & 63);
            if (group4(opsA, k)) break;
            if (k + 4 >= cnt) break;
            opsA = load_ops4(s_b, (k + 8) & 63);
            if (group4(opsB, k + 4)) break;
        }
        if (STORE && lane == 0) a.bw.bmask[range.x / BATCH + (uint32_t)(entered - 1)] = blended;
        wave_lds_fence();
    }

    // ---- outputs ---------------------------------------------------------------------------------
    const size_t HW = (size_t)a.grid.W * a.grid.H;
    if (inside) {
        const size_t pix = (size_t)pyi * a.grid.W + pxi;
        const float* __restrict__ bg = a.bg;
        const float Tf = T > 0.0f ? T : Tdead;
        a.out_color[pix] = Crg.x + Tf * bg[0];
        a.out_color[HW + pix] = Crg.y + Tf * bg[1];
        a.out_color[2 * HW + pix] = Cbd.x + Tf * bg[2];
        a.out_depth[pix] = Cbd.y;
        a.out_alpha[pix] = 1.0f - Tf;
    }
    if (STORE) {
        if (lane == 0) a.tw.fwd_exit[st] = make_uint2((uint32_t)n, (uint32_t)entered);
#ifdef EXA_PROBE_FWD
        if (lane == 0) {       // part_cnt is dead once the lists exist
            a.tw.part_cnt[4 * blockIdx.x] = (uint32_t)t0;
            a.tw.part_cnt[4 * blockIdx.x + 1] = (uint32_t)wall_clock64();
            // placement: HW_ID[15:0] (wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13) | XCC_ID << 16
            const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            a.tw.part_cnt[4 * blockIdx.x + 2] = (hw & 0xffffu) | ((xcc & 0xfu) << 16);
        }
#endif
        if (n > 0) {   // exit state -> the sub-tile's END slot (a fixed place the backward finds without `entered`)
            float* c = ckpt + (size_t)((n + BATCH - 1) / BATCH) * (5 * 64);
            c[0] = T > 0.0f ? T : Tdead; c[64] = Crg.x; c[128] = Crg.y; c[192] = Cbd.x; c[256] = Cbd.y;
        }
    }
}

static int max_subtiles(const RenderFwdArgs* a, int K) {
    int n = 0;
    for (int k = 0; k < K; ++k) n = max(n, a[k].grid.subtiles);
    return n;
}

hipError_t launch_sort_subtiles(const RenderFwdArgs* a, int K, hipStream_t s) {
    const int subtiles = max_subtiles(a, K);
    if (subtiles == 0) return hipSuccess;
    const int split_at = dev_knobs().split_subtiles;
    bool keep = false;                   // (all jobs of a call share the flag: the binding sets it per call)
    for (int k = 0; k < K; ++k) keep = keep || a[k].keep_sorted_keys != 0;
    const dim3 grid(subtiles + ORDER_WGS, K);
    if (subtiles >= split_at) {
        sort_short_kernel<<<dim3((subtiles + SBLOCK / 64 - 1) / (SBLOCK / 64), K), SBLOCK, 0, s>>>(make_batch(a, K));
        if (keep) sort_subtiles_kernel<true, true><<<grid, SBLOCK, 0, s>>>(make_batch(a, K));
        else sort_subtiles_kernel<true, false><<<grid, SBLOCK, 0, s>>>(make_batch(a, K));
    } else {
        if (keep) sort_subtiles_kernel<false, true><<<grid, SBLOCK, 0, s>>>(make_batch(a, K));
        else sort_subtiles_kernel<false, false><<<grid, SBLOCK, 0, s>>>(make_batch(a, K));
    }
    return hipGetLastError();
}

// (Round 3 measured where the waves of this launch go -- tools/gpu_fwd_timeline.py, probe build -- because an avatar view's
//  ~3 700 one-wave workgroups are all resident within a microsecond, i.e. the launch order is a static assignment of
//  lists to SIMDs made by the dispatcher: walked entries per SIMD mean 544, max 883; the last wave of a SIMD ends after
//  30.6 us on average, 38.6 us on the worst (= the launch).  A schedule that takes the assignment into the kernel -- one
//  16-wave workgroup per CU with a snake-dealt bundle of lists, its first sixteen dealt to the SIMDs the waves really sit
//  on (HW_ID), the rest pulled through an LDS counter -- evened the entries out (max 685) and was bit-identical, but
//  bought 0.4 us on C3 (the per-SIMD end times stopped following the entry counts: correlation 0.89 -> 0.61) and cost the
//  batched modes 5-7 % (8 330 -> 7 880 it/s at K = 8: four resident waves per SIMD instead of five, no overlap between the
//  jobs of a batch).  Snaking the launch order alone (periods 256 .. 2048): +-1 us.  Not kept.)
// All jobs of a batch share store_ctx (checked by the C ABI).  -DEXA_FWD_LDS_PAD=<bytes> (build-time probe) adds unused
// dynamic LDS per workgroup: fewer resident waves per CU, so that the tail of the length-sorted launch is dealt out
// dynamically as earlier waves retire instead of all sub-tiles being placed at once (measured: slower).
#ifndef EXA_FWD_LDS_PAD
#define EXA_FWD_LDS_PAD 0
#endif
hipError_t launch_render_fwd(const RenderFwdArgs* a, int K, hipStream_t s) {
    const int subtiles = max_subtiles(a, K);
    if (subtiles == 0) return hipSuccess;
    constexpr int pad = EXA_FWD_LDS_PAD;
    if (a[0].splats2) {                 // composite renders (all jobs of such a call are composites)
        if (a[0].store_ctx) render_fwd_kernel<true, true><<<dim3(subtiles, K), RBLOCK, (size_t)pad, s>>>(make_batch(a, K));
        else render_fwd_kernel<false, true><<<dim3(subtiles, K), RBLOCK, (size_t)pad, s>>>(make_batch(a, K));
    } else if (a[0].store_ctx)
        render_fwd_kernel<true, false><<<dim3(subtiles, K), RBLOCK, (size_t)pad, s>>>(make_batch(a, K));
    else
        render_fwd_kernel<false, false><<<dim3(subtiles, K), RBLOCK, (size_t)pad, s>>>(make_batch(a, K));
    return hipGetLastError();
}

}  // namespace exa
