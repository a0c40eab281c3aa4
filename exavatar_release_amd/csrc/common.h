// Shared device/host definitions of the gfx950 rasterizer (internal; the public ABI is
// include/exa_raster.h).
//
// Geometry of the pipeline (wave = 64 lanes):
//   sub-tile  = 8x8 pixels  = ONE wave (lane l -> pixel (l & 7, l >> 3)); all per-pixel kernels are
//               barrier-free, every wave owns its list, its LDS slice and its pixels.
//   cell      = 8x8 sub-tiles = 64x64 pixels; the unit of the first radix digit of the binning.
//   upstream's 16x16 tile only survives as the *semantic* clip rectangle of a Gaussian
//   (oracle step 7): a Gaussian is binned to the sub-tiles that lie inside its 16x16-tile rect AND
//   intersect the exact bounding box of {alpha >= 1/255}, which yields the same per-pixel result.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/exa_raster.h"

namespace exa {

constexpr int TILE = EXA_RASTER_TILE;      // upstream tile (clip rect granularity)
constexpr int SUB = 8;                     // sub-tile edge in pixels
constexpr int CELL_SUBS = 8;               // sub-tiles per cell edge
constexpr int CELL = SUB * CELL_SUBS;      // 64 px
constexpr int SUBS_PER_CELL = CELL_SUBS * CELL_SUBS;   // 64
constexpr int BLOCK = 256;                 // threads per workgroup (4 waves)
constexpr int CHUNK = 1024;                // Gaussians per workgroup in the per-Gaussian binning kernels
constexpr int MAX_CELLS = 4096;            // LDS histogram budget (48 KiB of counters) -> images up to 4096x4096
constexpr int HEADER_BYTES = 2560;     // ExaRasterHeader at 0, backward-order words at 128, launch-order cursors [8 regions][64 classes] at 256

// Batched launches: every kernel takes up to MAX_BATCH independent jobs (renders) by value in its kernel arguments and
// picks its own with blockIdx.y -- K views / K renders of one training iteration cost ONE launch per stage instead
// of K, and their workgroups fill the chip together (a single 1024x1024 view leaves most of the 256 CUs idle in
// every stage but the blend).  The job records live in the kernarg segment: indexing them with blockIdx.y compiles
// to scalar loads with a register offset (no scratch copy).
#ifndef EXA_MAX_BATCH
#define EXA_MAX_BATCH 8
#endif
constexpr int MAX_BATCH = EXA_MAX_BATCH;
template <typename T> struct Batch { T v[MAX_BATCH]; };
template <typename T>
inline Batch<T> make_batch(const T* a, int K) {          // unused slots repeat job 0 (never indexed: gridDim.y = K)
    Batch<T> b;
    for (int k = 0; k < MAX_BATCH; ++k) b.v[k] = a[k < K ? k : 0];
    return b;
}

constexpr float NEAR_CULL = 0.2f;
constexpr float LOWPASS = 0.3f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_EPS = 1e-4f;
constexpr float LOG2E = 1.4426950408889634f;

// 64-byte per-Gaussian splat record: one cache line per gather in the per-pixel kernels.
// Row 2 is what the blends stage as their colour operand AS IT IS (r, g, b, depth: one ds_write_b128 of the loaded row).
// Until round 4 the depth sat in row 0 and the staged float4 was assembled from two rows: the compiler placed that
// v_mov right behind the prefetch loads of the NEXT batch and a `s_waitcnt vmcnt(1)` with it -- the forward blend
// stalled for a whole gather round trip at the start of every 64-entry batch.
struct alignas(64) Splat {
    float px, py; uint32_t flags; int32_t radius;  // row 0: pixel centre, SH clamp bits (bit c: channel c clamped), radius
    float ca, cb, cc, opacity;                    // row 1: (-A/2, -B, -C/2) * log2(e) of the conic (A, B, C), opacity
    float r, g, b, depth;                         // row 2: colour, view-space depth
    uint32_t sub_x, sub_y, n_inst, inst_off;      // row 3: sx0 | sx1 << 16, sy0 | sy1 << 16 (sub-tile rect,
                                                  //        exclusive upper), instances, Gaussian-major offset
};
static_assert(sizeof(Splat) == 64, "Splat must be one 64-byte line");

// Per-instance partial gradients written (plain stores, no atomics) by render-backward: the sums over the 64 pixels
// of one sub-tile, for every instance the forward pass actually blended somewhere (its `touched` byte is set; the
// records of all other instances are never written and never read).  Indexed Gaussian-major:
// inst_off + (sy - sy0) * (sx1 - sx0) + (sx - sx0).  Ten floats in 16-byte rows:
//   row 0 = (sum s dx, sum s dy, sum s dx^2, sum s dx dy)        s = dL/dG * G
//   row 1 = (sum s dy^2, sum G dL/dalpha, sum w dL/dC_r, sum w dL/dC_g)    w = alpha T
//   row 2 = (sum w dL/dC_b, sum w dL/dDepth, -, -)
// PARTIAL_BYTES = 40: the ten floats packed (records on 8-byte boundaries, written as 16 + 16 + 8 bytes), since late round 5.
// Before: 48 (three whole rows; three separate arrays had dirtied twice as many 32-byte sectors, profiles/r02_hbm_traffic.md).
// Measured alternatives, interleaved runs on one box each (EXA_PARTIAL_BYTES keeps all three buildable):
//   64 (one record = one 64-byte line, written whole): C3 6 597 / 6 623 / 6 609 it/s against 6 646 / 6 637 / 6 650 with 48
//      (preprocess_bwd 19.6-20.0 vs 19.0 us by events): the extra 16 bytes per record cost more than the alignment gives;
//   40: 6 619 / 6 617 / 6 610 against 6 620 / 6 567 / 6 559 with 48 on one box (preprocess_bwd 19.4-19.6 vs 19.7-20.1 us,
//      render_bwd 43.9-44.5 vs 44.0-44.5), 6 603 / 6 600 / 6 588 / 6 605 against 6 603 / 6 606 / 6 603 / 6 610 on a second:
//      a wash in time (the two kernels are bound by instruction issue and latency, not by these bytes), gradients
//      bit-identical, a sixth less workspace.  By the counters (profiles/r05_hbm_traffic.md) render_bwd writes the SAME
//      38.8 MB -- a 40-byte record dirties two 32-byte sectors just as a 48-byte one does -- and preprocess_bwd fetches
//      21.7 instead of 23.6 MB.  Kept for the workspace.
#ifndef EXA_PARTIAL_BYTES
#define EXA_PARTIAL_BYTES 40
#endif
constexpr int PARTIAL_BYTES = EXA_PARTIAL_BYTES;
static_assert(PARTIAL_BYTES == 40 || PARTIAL_BYTES == 48 || PARTIAL_BYTES == 64, "partial records: ten floats packed, three rows, or one 64-byte line");
struct PartialWs { float4* rec; };
// rows of the record in slot `ps` (40-byte records sit on 8-byte boundaries: the accesses stay 16 bytes wide, which the
// hardware serves at dword alignment)
typedef float partial_v4 __attribute__((ext_vector_type(4), aligned(8)));
typedef float partial_v2 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ char* partial_at(float4* rec, size_t ps) { return reinterpret_cast<char*>(rec) + ps * PARTIAL_BYTES; }
__device__ __forceinline__ const char* partial_at(const float4* rec, size_t ps) { return reinterpret_cast<const char*>(rec) + ps * PARTIAL_BYTES; }
__device__ __forceinline__ void partial_load(const float4* rec, size_t ps, float4& q0, float4& q1, float4& q2) {
    const char* p = partial_at(rec, ps);
    const partial_v4 a = *reinterpret_cast<const partial_v4*>(p), b = *reinterpret_cast<const partial_v4*>(p + 16);
    q0 = make_float4(a.x, a.y, a.z, a.w); q1 = make_float4(b.x, b.y, b.z, b.w);
    if (PARTIAL_BYTES == 40) {
        const partial_v2 c = *reinterpret_cast<const partial_v2*>(p + 32);
        q2 = make_float4(c.x, c.y, 0.f, 0.f);
    } else {
        const partial_v4 c = *reinterpret_cast<const partial_v4*>(p + 32);
        q2 = make_float4(c.x, c.y, c.z, c.w);
    }
}

struct Grid {
    int W, H;
    int gx, gy;        // upstream 16x16 tile grid
    int sx, sy;        // sub-tile grid (8x8 px)
    int cx, cy, cells; // cell grid (64x64 px)
    int subtiles;      // cells * 64 (cell-major sub-tile index space, includes padding sub-tiles)
};
__host__ __device__ inline Grid make_grid(int W, int H) {
    Grid g;
    g.W = W; g.H = H;
    g.gx = (W + TILE - 1) / TILE;
    g.gy = (H + TILE - 1) / TILE;
    g.sx = (W + SUB - 1) / SUB;
    g.sy = (H + SUB - 1) / SUB;
    g.cx = (W + CELL - 1) / CELL;
    g.cy = (H + CELL - 1) / CELL;
    g.cells = g.cx * g.cy;
    g.subtiles = g.cells * SUBS_PER_CELL;
    return g;
}

__host__ __device__ inline uint64_t align256(uint64_t v) { return (v + 255) & ~uint64_t(255); }
__host__ __device__ inline int num_chunks(int P) { return (P + CHUNK - 1) / CHUNK; }

// Layout of the tile workspace (all sections 256-byte aligned).  Nothing in it needs zeroing by the caller or by
// a memset: every section is fully written by the kernel that produces it before anyone reads it.
#ifndef EXA_BIN_PARTS
#define EXA_BIN_PARTS 4
#endif
constexpr int BIN_PARTS = EXA_BIN_PARTS;   // workgroups per cell in the sub-tile binning (binning.hip)
struct TileWs {
    ExaRasterHeader* header;          // [1]          written by cell_scan_kernel
    uint32_t* cls_cur;                // [8][64]  launch-order slots handed out per (XCD region, list-length class) (zeroed by cell_scan)
    uint32_t* bwd_meta;               // [4]   launch order of the backward blend (render_fwd.hip order_slots writes it, in the
                                      //       sort launch: {batches in the main order, batches appended behind it, BWD_ORDER_MAGIC});
                                      //       the order itself lives in the bucket array of the bin workspace, dead by then
    unsigned long long* chunk_cell;   // [chunks][cells]  per-(chunk, cell) counts inst << 32 | entries, plain stores by
                                      //              preprocess (NO atomics: same-address device atomics of ~150 chunks
                                      //              serialise at the memory side); cell_scan replaces the low word by the
                                      //              chunk's first entry slot inside the cell bucket
    unsigned long long* cell_cnt;     // [cells]      column totals inst << 32 | entries
    uint2* cell_off;                  // [cells + 1]  exclusive prefix (entries, instances)
    uint32_t* chunk_inst;             // [chunks]     instances emitted by each chunk
    uint32_t* chunk_vis;              // [chunks]     Gaussians of the chunk that passed the culls
    uint32_t* chunk_tiles;            // [chunks]     16x16 tiles of the rects of the chunk's visible Gaussians (upstream's
                                      //              tiles_touched, summed: header.num_tile_instances)
    uint32_t* chunk_off;              // [chunks]     exclusive prefix of chunk_inst
    uint4* cell_desc;                 // [cells]      cells by descending instance count (heavy work first):
                                      //              {cell, first entry, end entry, first instance slot * 64} -- ONE load
                                      //              gives a per-cell workgroup its work
    uint2* ranges;                    // [subtiles]   [begin, end) into the instance arrays, cell-major
    uint4* slots;                     // [subtiles]   launch-order records {begin, end, st, 0}: ONE load gives a
                                      //              per-pixel-kernel workgroup everything it needs
    uint2* fwd_exit;                  // [subtiles]   {list length, batches the forward entered}
    uint32_t* cell_long;              // [cells]      by rank in cell_desc: number of the cell's lists longer than 64 keys
    uint32_t* part_cnt;               // [cells * BIN_PARTS][64]  entries per sub-tile counted by each workgroup of the
                                      //              two-launch sub-tile binning (a cell's parts are consecutive)
    uint4* part_desc;                 // [cells * BIN_PARTS]  work record of every such workgroup: {cell | rank << 12 |
                                      //              part << 24 | (parts - 1) << 28, first entry, end entry, first slot of the
                                      //              cell}; x = NO_PART for the workgroups beyond the sum (binning.hip)
    uint8_t* cls_code;                // [subtiles]   length_class of every list, written next to `ranges`: what the ordering
                                      //              workgroups of the sort launch histogram (16 lists per load, no indirection)
};
__host__ __device__ inline uint64_t tile_ws_bytes(int cells, int chunks) {
    return HEADER_BYTES + align256(uint64_t(chunks) * cells * 8) + align256(uint64_t(cells) * 8) +
           align256(uint64_t(cells + 1) * 8) + 4 * align256(uint64_t(chunks + 1) * 4) +
           align256(uint64_t(cells) * 16) + align256(uint64_t(cells) * SUBS_PER_CELL * 8) +
           align256(uint64_t(cells) * SUBS_PER_CELL * 16) + align256(uint64_t(cells) * SUBS_PER_CELL * 8) +
           align256(uint64_t(cells) * BIN_PARTS * SUBS_PER_CELL * 4) + align256(uint64_t(cells) * 4) +
           align256(uint64_t(cells) * BIN_PARTS * 16) + align256(uint64_t(cells) * SUBS_PER_CELL);
}
__host__ __device__ inline TileWs carve_tile_ws(void* base, int cells, int chunks) {
    char* p = static_cast<char*>(base);
    TileWs w;
    w.header = reinterpret_cast<ExaRasterHeader*>(p);
    w.cls_cur = reinterpret_cast<uint32_t*>(p + 256);
    w.bwd_meta = reinterpret_cast<uint32_t*>(p + 128);
    p += HEADER_BYTES;
    w.chunk_cell = reinterpret_cast<unsigned long long*>(p); p += align256(uint64_t(chunks) * cells * 8);
    w.cell_cnt = reinterpret_cast<unsigned long long*>(p); p += align256(uint64_t(cells) * 8);
    w.cell_off = reinterpret_cast<uint2*>(p); p += align256(uint64_t(cells + 1) * 8);
    w.chunk_inst = reinterpret_cast<uint32_t*>(p); p += align256(uint64_t(chunks + 1) * 4);
    w.chunk_vis = reinterpret_cast<uint32_t*>(p); p += align256(uint64_t(chunks + 1) * 4);
    w.chunk_tiles = reinterpret_cast<uint32_t*>(p); p += align256(uint64_t(chunks + 1) * 4);
    w.chunk_off = reinterpret_cast<uint32_t*>(p); p += align256(uint64_t(chunks + 1) * 4);
    w.cell_desc = reinterpret_cast<uint4*>(p); p += align256(uint64_t(cells) * 16);
    w.ranges = reinterpret_cast<uint2*>(p); p += align256(uint64_t(cells) * SUBS_PER_CELL * 8);
    w.slots = reinterpret_cast<uint4*>(p); p += align256(uint64_t(cells) * SUBS_PER_CELL * 16);
    w.fwd_exit = reinterpret_cast<uint2*>(p); p += align256(uint64_t(cells) * SUBS_PER_CELL * 8);
    w.part_cnt = reinterpret_cast<uint32_t*>(p); p += align256(uint64_t(cells) * BIN_PARTS * SUBS_PER_CELL * 4);
    w.cell_long = reinterpret_cast<uint32_t*>(p); p += align256(uint64_t(cells) * 4);
    w.part_desc = reinterpret_cast<uint4*>(p); p += align256(uint64_t(cells) * BIN_PARTS * 16);
    w.cls_code = reinterpret_cast<uint8_t*>(p);
    return w;
}

// Instance space.  Every non-empty sub-tile owns a 64-aligned range of BATCH slots (64 instances each):
// ceil(n / 64) batches + one end slot; `capacity` (always a multiple of 64) counts slots * 64, which is what
// header.num_rendered reports.  A batch slot is the unit of work of the backward pass (one wave each).
constexpr int BATCH = 64;
// bin workspace: keys[cap] (u64: depth bits << 32 | gaussian id), sorted ids[cap], cell buckets[cap]
// (16-byte entries {id, depth bits, sub-tile rect x, y} so the second digit reads them coalesced), then ONE
// contiguous section that cell_scatter_kernel zero-fills on every forward:
//   batch owner[cap / 64 + 1] ({sub-tile + 1, list begin, list length, 0}; all-zero = unused: the ONE load a backward
//   workgroup needs to find its work), blended mask[cap / 64 + 1] (u64 per batch slot, bit i = entry i of the batch
//   was blended into at least one pixel by the forward pass: the backward pass replays only those),
//   touched[cap] (u8 per instance, Gaussian-major: render-backward wrote this instance's partial sums),
// and per-pixel forward checkpoints ckpt[cap / 64 + 1][5][64] floats (T, Cr, Cg, Cb, depth at the START of every batch
// slot; the sub-tile's END slot holds the state at the forward's exit).
__host__ __device__ inline uint64_t bin_zero_bytes(uint64_t cap) {
    return align256((cap / BATCH + 1) * 16) + align256((cap / BATCH + 1) * 8) + align256(cap);
}
__host__ __device__ inline uint64_t bin_ws_bytes(uint64_t cap) {
    return align256(cap * 8) + align256(cap * 4) + align256(cap * 16) + bin_zero_bytes(cap) +
           align256((cap / BATCH + 1) * 5 * 64 * 4);
}
struct BinWs {
    unsigned long long* keys; uint32_t* sorted; uint4* bucket;
    uint4* owner; unsigned long long* bmask; uint8_t* touched;      // the zero-filled section starts at `owner`
    float* ckpt;
};
__host__ __device__ inline BinWs carve_bin_ws(void* base, uint64_t cap) {
    BinWs b;
    char* p = static_cast<char*>(base);
    b.keys = reinterpret_cast<unsigned long long*>(p); p += align256(cap * 8);
    b.sorted = reinterpret_cast<uint32_t*>(p); p += align256(cap * 4);
    b.bucket = reinterpret_cast<uint4*>(p); p += align256(cap * 16);
    b.owner = reinterpret_cast<uint4*>(p); p += align256((cap / BATCH + 1) * 16);
    b.bmask = reinterpret_cast<unsigned long long*>(p); p += align256((cap / BATCH + 1) * 8);
    b.touched = reinterpret_cast<uint8_t*>(p); p += align256(cap);
    b.ckpt = reinterpret_cast<float*>(p);
    return b;
}
// slots a cell needs at most: ceil(inst / 64) real batches + < 1 padding and 1 end slot per sub-tile
__host__ __device__ inline uint32_t cell_slots(uint32_t inst) {
    return inst ? (inst + BATCH - 1) / BATCH + 2 * SUBS_PER_CELL : 0u;
}

// backward scratch: one partial record per instance.
__host__ __device__ inline uint64_t grad_ws_bytes(uint64_t cap) { return align256(cap * PARTIAL_BYTES); }
// ... plus, behind them, the group scratch of a summed batch (preprocess_bwd.hip): 20 gradient floats + 3 statistics per Gaussian
constexpr int GROUP_SCRATCH_FLOATS = 23;
__host__ __device__ inline uint64_t group_scratch_bytes(uint64_t P) { return align256(P * GROUP_SCRATCH_FLOATS * sizeof(float)); }
__host__ __device__ inline PartialWs carve_grad_ws(void* base, uint64_t) {
    PartialWs w;
    w.rec = reinterpret_cast<float4*>(base);
    return w;
}

// Launch order of the forward blend: sub-tiles by DESCENDING list length (64 classes of 16 entries), empty ones
// last.  All non-empty sub-tiles are resident at once (~3.7 waves per SIMD) and the hardware deals consecutive
// workgroups round-robin over XCDs / CUs / SIMDs, so a length-sorted launch gives every SIMD one list of each
// size class instead of a random draw (measured on C3: the busiest SIMD had 2.06x the mean work with a
// cell-major order, and that SIMD set the kernel time).
constexpr int ORDER_CLASSES = 64;
// XCD region of a sub-tile (render_fwd.hip order_slots): `local` = its index inside the 8 x 8 sub-tiles of its cell.  A cell is
// cut into eight blocks of 2 x 4 sub-tiles (16 x 32 px), one per region, the same pattern in every cell: eight sub-tiles per
// region and cell exactly, so every region owns subtiles / 8 launch positions.  (Measured against 2 x 2 blocks dealt two per
// region and cell: C3 render_fwd FETCH 21.2 -> 18.1 MB, render_bwd 27.4 -> 25.7 MB, times equal; larger blocks cannot keep the
// per-cell balance.)
#ifndef EXA_XCD_REGIONS
#define EXA_XCD_REGIONS 8            // 1 (build-time, tools/build_variant.sh): ONE length-sorted stream, the order of rounds 2-5
#endif
constexpr int XCD_REGIONS = EXA_XCD_REGIONS;
static_assert(XCD_REGIONS == 8 || XCD_REGIONS == 1, "eight XCDs, or no regions at all");
__device__ __forceinline__ int xcd_region(int local) { return XCD_REGIONS == 8 ? ((local >> 1) & 3) | ((local >> 3) & 4) : 0; }
constexpr uint32_t BWD_ORDER_MAGIC = 0xB07DE7EDu;
constexpr int BWD_ORDER_DEPTH = 16;       // batches of a list that take part in the batch-major order (class 63 = 16 batches or more)
__device__ __forceinline__ int length_class(uint32_t n) { return n ? min(ORDER_CLASSES - 1, (int)((n + 15) / 16)) : 0; }

// cell-major sub-tile index st = cell * 64 + local  ->  coordinates
struct SubTile { int st, cell, local, ox, oy, gsx, gsy; };
__device__ __forceinline__ SubTile decode_subtile(int st, const Grid& g) {
    SubTile s;
    s.st = st;
    s.cell = st >> 6;
    s.local = st & 63;
    const int cxi = s.cell % g.cx, cyi = s.cell / g.cx;
    s.gsx = cxi * CELL_SUBS + (s.local & 7);        // global sub-tile coordinates
    s.gsy = cyi * CELL_SUBS + (s.local >> 3);
    s.ox = s.gsx * SUB;
    s.oy = s.gsy * SUB;
    return s;
}

// The per-(pixel, Gaussian) falloff shared by render-forward and render-backward, pinned (no
// compiler-chosen contraction) so both kernels take bit-identical skip decisions and the backward
// pass divides out exactly the alphas the forward pass multiplied in.  The conic is pre-scaled by
// log2(e): returns log2 of the Gaussian falloff;  G = exp2(power2).
__device__ __forceinline__ float gauss_power2(float A, float B, float C, float dx, float dy) {   // pre-scaled conic
#pragma clang fp contract(off)
    return __builtin_fmaf(A * dx, dx, __builtin_fmaf(C * dy, dy, (B * dx) * dy));
}
__device__ __forceinline__ float gauss_falloff2(float power2) { return __builtin_amdgcn_exp2f(power2); }

// ---- exact footprint arithmetic (binning.hip: which sub-tiles of a cell does a splat reach) ----
// The part of the ellipse  f(d) = A dx^2 + B dx dy + C dy^2 >= thr  (log2 domain, conic pre-scaled as in blend.h; A, C < 0,
// thr < 0; d = pixel - centre) inside a horizontal band dy in [yl, yh] is convex, so the pixel columns it reaches are exactly
// those that meet its x-extent [xa, xb].  On the line dy = y0 the ellipse spans  kA y0 -+ sqrt(X2 - D4 y0^2)  (kA = -B / 2A,
// X2 = thr / A, D4 = (4AC - B^2) / 4A^2); its rightmost (leftmost) point overall lies at dy = +ysr (-ysr), ysr = kC Xf with
// kC = -B / 2C and Xf the half extent in x; the right (left) end of the band's part is the line point at y0 = clamp(+ysr
// (-ysr), yl, yh).  Conservative: thr is relaxed by 1e-3 + 1e-5 * (a bound of the term magnitudes) to cover the rounding of
// the per-pixel evaluation, the callers relax the interval by 1e-3 (1 + |x|) px for the arithmetic here (tools/footprint_check.py
// compares this arithmetic with the per-pixel rule by brute force); NaN / degenerate conics keep everything.
struct Footprint { float kA, ysr, X2, D4; bool test; };
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
// (A, B, C, opacity) = row 1 of the splat record; xm, ym: the largest |pixel - centre| inside the region the caller asks about
__device__ __forceinline__ Footprint make_footprint(float A, float B, float C, float opacity, float xm, float ym) {
    Footprint f;
    const float rA = __builtin_amdgcn_rcpf(A), rC = __builtin_amdgcn_rcpf(C);
    const float As = A - 0.25f * B * B * rC;                    // f along the line of the x-extreme points: As dx^2
    f.test = A < 0.f && C < 0.f && As < 0.f;                    // anything else (never for a visible splat): keep everything
    const float mag = fabsf(A) * xm * xm + fabsf(B) * xm * ym + fabsf(C) * ym * ym;
    const float thr = -__log2f(255.0f * opacity) - 1e-3f - 1e-5f * mag;   // alpha >= 1/255 <=> f >= -log2(255 o)
    f.kA = -0.5f * B * rA;
    f.ysr = -0.5f * B * rC * __builtin_amdgcn_sqrtf(thr * __builtin_amdgcn_rcpf(As));
    f.X2 = thr * rA;
    f.D4 = C * As * rA * rA;
    return f;
}
// x-extent of the footprint inside the band [yl, yh]; false: the band misses it
__device__ __forceinline__ bool footprint_band(const Footprint& f, float yl, float yh, float& xa, float& xb) {
    const float yR = clampf(f.ysr, yl, yh), yL = clampf(-f.ysr, yl, yh);
    const float hR2 = f.X2 - f.D4 * yR * yR, hL2 = f.X2 - f.D4 * yL * yL;
    if (hR2 < 0.f || hL2 < 0.f) return false;
    xb = f.kA * yR + __builtin_amdgcn_sqrtf(hR2);
    xa = f.kA * yL - __builtin_amdgcn_sqrtf(hL2);
    return true;
}
// Wave-private LDS hand-off (one wave writes, the same wave reads other lanes' data): LDS operations of
// one wave execute in order; this stops the compiler from reordering across it and drains the LDS
// counter ONLY (no vmcnt wait, so global prefetches stay in flight across it).
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// Host-side launch helpers implemented one per .hip file.  Every launcher takes K <= MAX_BATCH jobs.
struct PreprocessArgs {
    int P, sh_M, sh_degree;
    Grid grid;
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    const float* viewmatrix; const float* projmatrix; const float* campos;
    const float* means3D; const float* shs; const float* colors_precomp; const float* opacities;
    const float* scales; const float* rotations; const float* cov3D_precomp;
    int32_t* radii; uint8_t* is_vis; Splat* splats; TileWs tw;
};
hipError_t launch_preprocess_fwd(const PreprocessArgs* a, int K, hipStream_t s);
hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);
struct Proj16 { float m[16]; };
hipError_t launch_camera_block(const float* R, const float* t, const Proj16& proj, float* view_out, float* proj_out,
                               float* campos_out, const float* focal, float fx, float fy, uint32_t* host_flag, uint32_t tag,
                               hipStream_t s);

hipError_t launch_zero(void* p, size_t bytes, hipStream_t s);

// cell_scatter_kernel can do the scans of the count matrix itself (no col_scan / cell_scan launches): worth it while
// the matrix is small enough for every scatter workgroup to re-read it from L2
constexpr int MERGE_MAX_CELLS = 1024, MERGE_MAX_CHUNKS = 1024, MERGE_MAX_MATRIX = 1 << 16;
struct BinArgs {           // one job of the binning stages (binning.hip)
    int P, chunks, merged;
    Grid grid;
    Splat* splats; TileWs tw; BinWs bw; uint64_t capacity;
    uint32_t* host_hdr; uint32_t hdr_tag;        // ExaRasterForwardJob.host_header / header_tag (NULL = off)
};
hipError_t launch_cell_scan(const BinArgs* a, int K, hipStream_t s);
hipError_t launch_cell_scatter(const BinArgs* a, int K, hipStream_t s);
hipError_t launch_subtile_bin(const BinArgs* a, int K, hipStream_t s);

// Composite renders (compose.hip): the sorted list of a sub-tile is the MERGE of the sorted lists of two finished renders
// A and B of the same camera; an id carries its source in bit 31 (set = B).  `splats2` != nullptr selects that decoding.
constexpr uint32_t SRC_B = 0x80000000u;
struct RenderFwdArgs {
    Grid grid;
    const Splat* splats; TileWs tw; BinWs bw; uint64_t capacity;
    const float* bg; float* out_color; float* out_depth; float* out_alpha; int store_ctx;
    int keep_sorted_keys;      // sort_subtiles: also write the sorted 64-bit keys back over bw.keys (merge source)
    const Splat* splats2;      // composite: records of source B (ids with SRC_B); `splats` = source A
    // composite, optional: source A's finished images + its background.  Equal backgrounds => a sub-tile without B entries
    // has an empty list (compose.hip) and its pixels are copied from A's images (reuse_a_pixels below)
    const float* src_color; const float* src_depth; const float* src_alpha; const float* src_bg;
};
// Composite renders: may the pixels of source A stand in for sub-tiles without B entries?  (Wave-uniform: six scalar loads.)
__device__ __forceinline__ bool reuse_a_pixels(const float* src_color, const float* src_bg, const float* bg) {
    return src_color != nullptr && src_bg[0] == bg[0] && src_bg[1] == bg[1] && src_bg[2] == bg[2];
}
hipError_t launch_sort_subtiles(const RenderFwdArgs* a, int K, hipStream_t s);
// Composite of two finished renders (compose.hip): ranges / header / launch order / zero-fill, then the merged id lists
struct ComposeArgs {
    Grid grid;
    TileWs tw_a, tw_b; BinWs bw_a, bw_b;       // sources (sorted keys in bw.keys: forward with keep_sorted_keys)
    TileWs tw; BinWs bw; uint64_t capacity, capacity_b;   // the composite's own: bw.touched holds capacity_b bytes
    uint32_t* host_hdr; uint32_t hdr_tag;
    const float* src_color; const float* src_bg; const float* bg;    // see RenderFwdArgs.src_color
    int P_a, P_b;                                                     // optional outputs radii / is_vis of cat(A, B)
    const int32_t* radii_a; const int32_t* radii_b; int32_t* radii_out;
    const uint8_t* vis_a; const uint8_t* vis_b; uint8_t* vis_out;
};
hipError_t launch_compose(const ComposeArgs* a, int K, hipStream_t s, int parts = 3);
// bin workspace of a composite: merged ids [cap] | zero-filled: owner [cap / 64 + 1], blended mask [cap / 64 + 1], touched
// [capacity_b] (B's Gaussian-major instance numbering) | checkpoints [cap / 64 + 1][5][64]
__host__ __device__ inline uint64_t compose_zero_bytes(uint64_t cap, uint64_t cap_b) {
    return align256((cap / BATCH + 1) * 16) + align256((cap / BATCH + 1) * 8) + align256(cap_b);
}
__host__ __device__ inline uint64_t compose_bin_bytes(uint64_t cap, uint64_t cap_b) {
    return align256(cap * 4) + compose_zero_bytes(cap, cap_b) + align256((cap / BATCH + 1) * 5 * 64 * 4);
}
__host__ __device__ inline BinWs carve_compose_ws(void* base, uint64_t cap, uint64_t cap_b) {
    BinWs b;
    char* p = static_cast<char*>(base);
    b.keys = nullptr; b.bucket = nullptr;
    b.sorted = reinterpret_cast<uint32_t*>(p); p += align256(cap * 4);
    b.owner = reinterpret_cast<uint4*>(p); p += align256((cap / BATCH + 1) * 16);
    b.bmask = reinterpret_cast<unsigned long long*>(p); p += align256((cap / BATCH + 1) * 8);
    b.touched = reinterpret_cast<uint8_t*>(p); p += align256(cap_b);
    b.ckpt = reinterpret_cast<float*>(p);
    return b;
}
hipError_t launch_render_fwd(const RenderFwdArgs* a, int K, hipStream_t s);

struct RenderBwdArgs {
    Grid grid; uint64_t capacity; int P;
    const Splat* splats; TileWs tw; BinWs bw; const float* bg;
    const float* dL_dcolor; const float* dL_ddepth; const float* dL_dalpha;
    const float* const* dL_dcolor_ind;   // optional: the colour gradient's address is loaded from here at execution time
    uint32_t used_slots;                 // batch slots the forward used (0 = capacity / 64): bounds the launch, nothing else
    PartialWs partials;
    int grad_first;        // Gaussians below this index are constants (ExaRasterBackwardJob.grad_first)
    const Splat* splats2;  // composite (PREFIX instantiation): records of source B = the trainable Gaussians (ids with SRC_B),
    int P2;                // `splats` / `P` = source A = constants; partial slots in B's Gaussian-major numbering
};
hipError_t launch_render_bwd(const RenderBwdArgs* a, int K, hipStream_t s);

struct PreprocessBwdArgs {
    int P, sh_M, sh_degree;
    Grid grid;
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    const float* viewmatrix; const float* projmatrix; const float* campos;
    const float* means3D; const float* shs; const float* opacities;
    const float* scales; const float* rotations; const float* cov3D_precomp;
    const int32_t* radii; const Splat* splats; PartialWs partials; const uint8_t* touched;
    uint64_t partial_bytes;                                    // size of partials.rec (PARTIAL_BYTES x capacity): bound of the buffer loads
    const ExaRasterHeader* header;
    float* dL_dmeans2D; float* dL_dmeans3D; float* dL_dcolors; float* dL_dopacity;
    float* dL_dscales; float* dL_drotations; float* dL_dsh; float* dL_dcov3D;
    float* dens_accum; float* dens_cnt; float* dens_rmax;       // optional fused densification statistics (per view)
    int grad_first;        // Gaussians below this index are constants: no work, no output rows (output row = idx - grad_first)
    int accumulate;        // != 0: the per-Gaussian outputs (all but dL_dmeans2D) hold values this call adds to
    float* group_scratch;  // GROUP_SCRATCH_FLOATS x P floats behind this job's partial records (grad workspace): where the second
    //                        workgroup row of a summed batch of more than four views leaves its sum (preprocess_bwd.hip)
};
// sum_shared != 0: the K jobs are K views of the SAME Gaussians (identical input pointers and P): one thread
// per Gaussian walks the K views and writes the SUM of their gradients to job 0's outputs (dL_dmeans2D stays per view).
// dens_shared != 0 (with sum_shared): the K views accumulate into job 0's densification statistics (one write per Gaussian).
hipError_t launch_preprocess_bwd(const PreprocessBwdArgs* a, int K, int sum_shared, int dens_shared, hipStream_t s);

// The three RUN-TIME developer knobs of the library (environment, read once; api.hip): they exist because tests have to reach
// code paths that only large images take (tests/test_gpu_parity.py) and switch the exact-footprint test off for an A/B.
// Every other variant ever measured (render_bwd chunk size / slots per wave / waves per workgroup, LDS occupancy pads,
// partial-record size, gather chunk) is a COMPILE-TIME macro: tools/build_variant.sh <rev> <name> -DEXA_...=...
constexpr int SINGLE_PART_CELLS = 1024;      // default of DevKnobs.single_cells (2048 x 2048 px)
constexpr int SPLIT_SORT_SUBTILES = 65536;   // default of DevKnobs.split_subtiles
struct DevKnobs {
    bool footprint;          // EXA_FOOTPRINT=0: sub-tile binning keeps the whole bounding rect of every splat
    int single_cells;        // EXA_BIN_SINGLE_CELLS=<n>: images with at least n cells take the one-workgroup-per-cell binning
    int split_subtiles;      // EXA_SORT_SPLIT_SUBTILES=<n>: images with at least n sub-tiles sort short lists in their own launch
};
const DevKnobs& dev_knobs();
hipError_t launch_ssim_fwd(int N, int H, int W, const float* img1, const float* img2, float* map, float* dm_dmu1,
                           float* dm_dE11, float* dm_dE12, hipStream_t s);
hipError_t launch_ssim_bwd(int N, int H, int W, const float* img1, const float* img2, const float* dL_dmap,
                           const float* dm_dmu1, const float* dm_dE11, const float* dm_dE12, float* dL_dimg1,
                           hipStream_t s);
hipError_t launch_photo_loss(int B, int C, int H, int W, const int* crop, const float* x, const float* y, const float* l1w,
                             const float* smask, float w_l1, float w_ssim, float* maps, float* partials, float* dL_dx,
                             int stage, hipStream_t s);
hipError_t launch_l1(int B, int C, int H, int W, const int* crop, const float* x, const float* y, const float* mask,
                     const float* bg, const float* g, float* out, int backward, hipStream_t s);
hipError_t launch_densify_stats(int P, const float* g2d, const int32_t* radii, float* accum, float* cnt, float* rmax,
                                hipStream_t s);

}  // namespace exa
