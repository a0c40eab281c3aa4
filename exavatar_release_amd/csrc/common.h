// Shared device/host definitions of the gfx950 rasterizer (internal; the public ABI is
// include/exa_raster.h).  Wave = 64 lanes, workgroup = 256 threads = one 16x16 pixel tile,
// wave w of a tile owns the 16x4 pixel strip of rows 4w..4w+3 (64-byte image row segments).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/exa_raster.h"

namespace exa {

constexpr int TILE = EXA_RASTER_TILE;
constexpr int BLOCK = 256;          // threads per workgroup
constexpr int NSUB = 8;             // sub-counters per tile (spreads same-address atomic contention)
constexpr int SORT_CAP = 4096;      // per-tile list length sorted inside LDS (32 KiB of keys)
constexpr int HEADER_BYTES = 256;

constexpr float NEAR_CULL = 0.2f;
constexpr float LOWPASS = 0.3f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_EPS = 1e-4f;

// 64-byte per-Gaussian splat record: one cache line per gather in the per-tile kernels.
struct alignas(64) Splat {
    float px, py, depth; int32_t radius;          // row 0
    float ca, cb, cc, opacity;                    // row 1: conic (A, B, C) + opacity
    float r, g, b; uint32_t flags;                // row 2: colour + SH clamp bits (bit c: channel c clamped)
    uint32_t rect_x, rect_y, tiles, pad;          // row 3: x0 | x1 << 16, y0 | y1 << 16, tiles touched
};
static_assert(sizeof(Splat) == 64, "Splat must be one 64-byte line");

// Per-Gaussian screen-space gradient accumulator written by render-backward (atomics), 64 B.
struct alignas(64) GradAcc {
    float dpx, dpy;          // dL/d(pixel centre), pixel units
    float dA, dB, dC;        // dL/d(conic)
    float dop;               // dL/d(opacity)
    float dr, dg, db;        // dL/d(colour)
    float dz;                // dL/d(view-space depth)
    float pad[6];
};
static_assert(sizeof(GradAcc) == 64, "GradAcc must be one 64-byte line");

struct Grid {
    int W, H, gx, gy, tiles;
};
__host__ __device__ inline Grid make_grid(int W, int H) {
    Grid g;
    g.W = W; g.H = H;
    g.gx = (W + TILE - 1) / TILE;
    g.gy = (H + TILE - 1) / TILE;
    g.tiles = g.gx * g.gy;
    return g;
}

// Layout of the tile workspace (all sections 256-byte aligned).
struct TileWs {
    ExaRasterHeader* header;     // [1] (+ padding to HEADER_BYTES)
    uint32_t* counts;            // [NSUB][tiles]   instances per (sub-counter, tile)
    uint32_t* cursor;            // [NSUB][tiles]   exclusive prefix, advanced by the scatter pass
    uint2* ranges;               // [tiles]         [begin, end) into the instance arrays
    uint32_t* max_contrib;       // [tiles]         last list position any pixel of the tile blended
};
__host__ __device__ inline uint64_t align256(uint64_t v) { return (v + 255) & ~uint64_t(255); }
__host__ __device__ inline uint64_t tile_ws_bytes(int tiles) {
    return HEADER_BYTES + 2 * align256(uint64_t(NSUB) * tiles * 4) + align256(uint64_t(tiles) * 8) +
           align256(uint64_t(tiles) * 4);
}
__host__ __device__ inline TileWs carve_tile_ws(void* base, int tiles) {
    char* p = static_cast<char*>(base);
    TileWs w;
    w.header = reinterpret_cast<ExaRasterHeader*>(p); p += HEADER_BYTES;
    w.counts = reinterpret_cast<uint32_t*>(p); p += align256(uint64_t(NSUB) * tiles * 4);
    w.cursor = reinterpret_cast<uint32_t*>(p); p += align256(uint64_t(NSUB) * tiles * 4);
    w.ranges = reinterpret_cast<uint2*>(p); p += align256(uint64_t(tiles) * 8);
    w.max_contrib = reinterpret_cast<uint32_t*>(p);
    return w;
}

// bin workspace: keys[capacity] (u64: depth bits << 32 | gaussian id), then sorted ids[capacity].
__host__ __device__ inline uint64_t bin_ws_bytes(uint64_t cap) { return align256(cap * 8) + align256(cap * 4); }
struct BinWs { unsigned long long* keys; uint32_t* sorted; };
__host__ __device__ inline BinWs carve_bin_ws(void* base, uint64_t cap) {
    BinWs b;
    b.keys = static_cast<unsigned long long*>(base);
    b.sorted = reinterpret_cast<uint32_t*>(static_cast<char*>(base) + align256(cap * 8));
    return b;
}

// image workspace: final_T[H*W] f32, n_contrib[H*W] u32.
__host__ __device__ inline uint64_t img_ws_bytes(int W, int H) { return 2 * align256(uint64_t(W) * H * 4); }
struct ImgWs { float* final_T; uint32_t* n_contrib; };
__host__ __device__ inline ImgWs carve_img_ws(void* base, int W, int H) {
    ImgWs i;
    i.final_T = static_cast<float*>(base);
    i.n_contrib = reinterpret_cast<uint32_t*>(static_cast<char*>(base) + align256(uint64_t(W) * H * 4));
    return i;
}

// Which sub-counter a Gaussian uses: constant per 256-Gaussian block, identical in count and scatter.
__host__ __device__ inline int sub_of(int gaussian) { return (gaussian >> 8) & (NSUB - 1); }

// The two per-(pixel, Gaussian) expressions shared by render-forward and render-backward.  They are
// pinned (no compiler-chosen contraction) so both kernels take bit-identical skip decisions and the
// backward pass divides out exactly the alphas the forward pass multiplied in.
__device__ __forceinline__ float gauss_power(float A, float B, float C, float dx, float dy) {
#pragma clang fp contract(off)
    const float q = __builtin_fmaf(C * dy, dy, (A * dx) * dx);
    return __builtin_fmaf(-(B * dx), dy, -0.5f * q);
}
__device__ __forceinline__ float gauss_falloff(float power) { return __expf(power); }

// Host-side launch helpers implemented one per .hip file.
struct PreprocessArgs {
    int P, sh_M, sh_degree;
    Grid grid;
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    const float* viewmatrix; const float* projmatrix; const float* campos;
    const float* means3D; const float* shs; const float* colors_precomp; const float* opacities;
    const float* scales; const float* rotations; const float* cov3D_precomp;
    int32_t* radii; Splat* splats; uint32_t* counts; ExaRasterHeader* header;
};
hipError_t launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t s);
hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);

hipError_t launch_zero(void* p, size_t bytes, hipStream_t s);
hipError_t launch_tile_scan(const TileWs& w, int tiles, hipStream_t s);
hipError_t launch_scatter(int P, const Splat* splats, const TileWs& w, int tiles, int gx, const BinWs& b,
                          uint64_t capacity, hipStream_t s);

struct RenderFwdArgs {
    Grid grid;
    const Splat* splats; TileWs tw; BinWs bw; uint64_t capacity; ImgWs iw;
    const float* bg; float* out_color; float* out_depth; float* out_alpha; int store_ctx;
};
hipError_t launch_render_fwd(const RenderFwdArgs& a, hipStream_t s);

struct RenderBwdArgs {
    Grid grid;
    const Splat* splats; TileWs tw; BinWs bw; ImgWs iw; const float* bg;
    const float* dL_dcolor; const float* dL_ddepth; const float* dL_dalpha;
    GradAcc* acc;
};
hipError_t launch_render_bwd(const RenderBwdArgs& a, hipStream_t s);

struct PreprocessBwdArgs {
    int P, sh_M, sh_degree;
    Grid grid;
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    const float* viewmatrix; const float* projmatrix; const float* campos;
    const float* means3D; const float* shs; const float* opacities;
    const float* scales; const float* rotations; const float* cov3D_precomp;
    const int32_t* radii; const Splat* splats; const GradAcc* acc;
    float* dL_dmeans2D; float* dL_dmeans3D; float* dL_dcolors; float* dL_dopacity;
    float* dL_dscales; float* dL_drotations; float* dL_dsh; float* dL_dcov3D;
};
hipError_t launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t s);

}  // namespace exa
