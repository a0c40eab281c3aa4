// C ABI of the rasterizer (include/exa_raster.h): argument validation, workspace carving, kernel
// sequencing.  No torch types, no allocation, no hidden synchronisation (unless settings->debug).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

using namespace exa;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* what) {
    snprintf(g_err, sizeof(g_err), "exa_raster: %s", what);
    return code;
}
int fail_hip(hipError_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "exa_raster: HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), where);
    return (int)e;
}

#define EXA_HIP(expr, where)                                   \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return fail_hip(e_, where);      \
    } while (0)

int debug_sync(const ExaRasterSettings* s, hipStream_t st, const char* where) {
    if (!s->debug) return 0;
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, where);
    return 0;
}

enum { K_ZERO, K_PREPROCESS_FWD, K_CELL_SCAN, K_CELL_SCATTER, K_SUBTILE_BIN, K_SORT, K_RENDER_FWD, K_RENDER_BWD,
       K_PREPROCESS_BWD, K_COUNT };
static_assert(K_COUNT == EXA_RASTER_TIMING_SLOTS, "timing slots");
const char* const k_names[K_COUNT] = {"zero", "preprocess_fwd", "cell_scan", "cell_scatter", "subtile_bin",
                                      "sort_subtiles", "render_fwd", "render_bwd", "preprocess_bwd"};
struct Timing {
    bool enabled = false, created = false;
    hipEvent_t ev[K_COUNT][2];
    bool used[K_COUNT] = {};
};
Timing g_t;   // process-wide on purpose: autograd runs backward on its own thread

// Enqueue `expr` (a hipError_t expression) bracketed by timing events when timing is on.
#define EXA_TIMED(slot, expr, where)                                        \
    do {                                                                    \
        if (g_t.enabled) EXA_HIP(hipEventRecord(g_t.ev[slot][0], st), where); \
        EXA_HIP((expr), where);                                             \
        if (g_t.enabled) {                                                  \
            EXA_HIP(hipEventRecord(g_t.ev[slot][1], st), where);            \
            g_t.used[slot] = true;                                          \
        }                                                                   \
    } while (0)

int check_settings(const ExaRasterSettings* s) {
    if (!s) return fail(EXA_RASTER_E_NULLPTR, "settings is NULL");
    if (s->image_height < 0 || s->image_width < 0) return fail(EXA_RASTER_E_INVALID, "negative image size");
    if (make_grid(s->image_width, s->image_height).cells > MAX_CELLS || s->image_width > 32768 || s->image_height > 32768)
        return fail(EXA_RASTER_E_INVALID, "image too large (more than 4096 cells of 64x64 pixels)");
    if (!s->bg || !s->viewmatrix || !s->projmatrix || !s->campos)
        return fail(EXA_RASTER_E_NULLPTR, "settings bg/viewmatrix/projmatrix/campos must be device pointers");
    if (!(s->tanfovx > 0.f) || !(s->tanfovy > 0.f)) return fail(EXA_RASTER_E_INVALID, "tanfov must be > 0");
    if (s->sh_degree < 0 || s->sh_degree > 3) return fail(EXA_RASTER_E_INVALID, "sh_degree must be 0..3");
    return 0;
}

int check_inputs(int32_t P, int32_t sh_M, const float* means3D, const float* shs, const float* colors_precomp,
                 const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                 int sh_degree) {
    if (P < 0) return fail(EXA_RASTER_E_INVALID, "P < 0");
    if (P == 0) return 0;
    if (P > (1 << 26) - 4) return fail(EXA_RASTER_E_INVALID, "more than 2^26 - 4 Gaussians (32-bit byte offsets of the 64-byte splat records)");
    if (!means3D || !opacities) return fail(EXA_RASTER_E_NULLPTR, "means3D / opacities is NULL");
    if ((shs == nullptr) == (colors_precomp == nullptr))
        return fail(EXA_RASTER_E_INVALID, "provide exactly one of shs / colors_precomp");
    const bool has_sr = scales != nullptr && rotations != nullptr;
    if ((scales != nullptr) != (rotations != nullptr) || has_sr == (cov3D_precomp != nullptr))
        return fail(EXA_RASTER_E_INVALID, "provide exactly one of (scales + rotations) / cov3D_precomp");
    if (shs && sh_M < (sh_degree + 1) * (sh_degree + 1))
        return fail(EXA_RASTER_E_INVALID, "sh_M smaller than (sh_degree + 1)^2");
    return 0;
}

}  // namespace

// The library's only environment reads (common.h, DevKnobs): developer knobs, read once.
const exa::DevKnobs& exa::dev_knobs() {
    static const DevKnobs k = [] {
        DevKnobs d;
        const char* e = getenv("EXA_FOOTPRINT");
        d.footprint = !e || atoi(e) != 0;
        e = getenv("EXA_BIN_SINGLE_CELLS");
        d.single_cells = e ? atoi(e) : SINGLE_PART_CELLS;
        e = getenv("EXA_SORT_SPLIT_SUBTILES");
        d.split_subtiles = e ? atoi(e) : SPLIT_SORT_SUBTILES;
        return d;
    }();
    return k;
}

extern "C" {

int exa_raster_version(void) { return EXA_RASTER_VERSION; }

const char* exa_raster_last_error(void) { return g_err; }

int exa_raster_workspace_sizes(int32_t P, int32_t W, int32_t H, uint64_t capacity, ExaRasterWorkspaceSizes* out) {
    if (!out) return fail(EXA_RASTER_E_NULLPTR, "out is NULL");
    if (P < 0 || W < 0 || H < 0) return fail(EXA_RASTER_E_INVALID, "negative size");
    const Grid g = make_grid(W, H);
    out->geom_bytes = align256(uint64_t(P) * sizeof(Splat));
    out->tile_bytes = tile_ws_bytes(g.cells, num_chunks(P));
    out->bin_bytes = bin_ws_bytes(capacity);
    out->grad_bytes = grad_ws_bytes(capacity) + group_scratch_bytes((uint64_t)P);
    return 0;
}

}  // extern "C"

namespace {

int check_forward_job(const ExaRasterForwardJob& j, bool stage1, bool stage2) {
    int rc = check_settings(j.settings);
    if (rc) return rc;
    if (stage1) {
        rc = check_inputs(j.P, j.sh_M, j.means3D, j.shs, j.colors_precomp, j.opacities, j.scales, j.rotations,
                          j.cov3D_precomp, j.settings->sh_degree);
        if (rc) return rc;
        if (j.P > 0 && !j.radii) return fail(EXA_RASTER_E_WORKSPACE, "workspace / radii is NULL");
    }
    if (j.P < 0) return fail(EXA_RASTER_E_INVALID, "P < 0");
    if (!j.tile_ws || (j.P > 0 && !j.geom_ws)) return fail(EXA_RASTER_E_WORKSPACE, "workspace / radii is NULL");
    if (stage2) {
        if (j.capacity > 0 && !j.bin_ws) return fail(EXA_RASTER_E_WORKSPACE, "workspace is NULL");
        if (j.capacity % BATCH) return fail(EXA_RASTER_E_INVALID, "capacity must be a multiple of 64");
        if (!j.out_color || !j.out_depth || !j.out_alpha) return fail(EXA_RASTER_E_NULLPTR, "output image is NULL");
    }
    return 0;
}

BinArgs bin_args(const ExaRasterForwardJob& j) {
    BinArgs b;
    b.P = j.P; b.chunks = num_chunks(j.P); b.merged = 0;
    b.grid = make_grid(j.settings->image_width, j.settings->image_height);
    b.splats = static_cast<Splat*>(j.geom_ws);
    b.tw = carve_tile_ws(j.tile_ws, b.grid.cells, b.chunks);
    b.bw = carve_bin_ws(j.bin_ws, j.capacity);
    b.capacity = j.capacity;
    b.host_hdr = static_cast<uint32_t*>(j.host_header); b.hdr_tag = j.header_tag;
    return b;
}

// stage 1 of up to MAX_BATCH jobs: one launch per kernel
// the scans can move into cell_scatter_kernel when every job's count matrix is small (fused stage 1 + 2 calls only:
// the two-stage protocol needs the header on the host before stage 2 is launched)
bool can_merge_scans(const ExaRasterForwardJob* jobs, int K) {
    for (int k = 0; k < K; ++k) {
        const int cells = make_grid(jobs[k].settings->image_width, jobs[k].settings->image_height).cells;
        const int chunks = num_chunks(jobs[k].P);
        if (cells == 0 || cells > MERGE_MAX_CELLS || chunks > MERGE_MAX_CHUNKS || (int64_t)cells * chunks > MERGE_MAX_MATRIX)
            return false;
    }
    return true;
}

int forward_bin_group(const ExaRasterForwardJob* jobs, int K, hipStream_t st, bool skip_scans = false) {
    PreprocessArgs pa[MAX_BATCH];
    BinArgs ba[MAX_BATCH];
    const ExaRasterSettings* s0 = jobs[0].settings;
    for (int k = 0; k < K; ++k) {
        const ExaRasterForwardJob& j = jobs[k];
        const ExaRasterSettings* s = j.settings;
        PreprocessArgs& a = pa[k];
        a.P = j.P; a.sh_M = j.sh_M; a.sh_degree = s->sh_degree;
        a.grid = make_grid(s->image_width, s->image_height);
        a.tanfovx = s->tanfovx; a.tanfovy = s->tanfovy;
        a.focal_x = (float)s->image_width / (2.0f * s->tanfovx);
        a.focal_y = (float)s->image_height / (2.0f * s->tanfovy);
        a.scale_modifier = s->scale_modifier;
        a.viewmatrix = s->viewmatrix; a.projmatrix = s->projmatrix; a.campos = s->campos;
        a.means3D = j.means3D; a.shs = j.shs; a.colors_precomp = j.colors_precomp; a.opacities = j.opacities;
        a.scales = j.scales; a.rotations = j.rotations; a.cov3D_precomp = j.cov3D_precomp;
        a.radii = j.radii; a.is_vis = j.is_vis; a.splats = static_cast<Splat*>(j.geom_ws);
        a.tw = carve_tile_ws(j.tile_ws, a.grid.cells, num_chunks(j.P));
        ba[k] = bin_args(j);
    }
    int rc;
    EXA_TIMED(K_PREPROCESS_FWD, launch_preprocess_fwd(pa, K, st), "preprocess_fwd");
    if ((rc = debug_sync(s0, st, "preprocess_fwd"))) return rc;
    if (skip_scans) return 0;
    EXA_TIMED(K_CELL_SCAN, launch_cell_scan(ba, K, st), "cell_scan");
    if ((rc = debug_sync(s0, st, "cell_scan"))) return rc;
    return 0;
}

// `store_ctx` of the batched forward calls: bit 0 = keep the backward context; the stage bits split a call in two so that a
// caller can put other work between the sorted lists and the blend (include/exa_raster.h, EXA_RASTER_STAGE_*)
inline bool stage_lists(int f) { return !(f & (EXA_RASTER_STAGE_BLEND_ONLY | EXA_RASTER_STAGE_SORT_ONLY)); }                // up to the unsorted lists
inline bool stage_sort(int f) { return !(f & (EXA_RASTER_STAGE_BLEND_ONLY | EXA_RASTER_STAGE_NO_SORT)); }
inline bool stage_blend(int f) { return !(f & (EXA_RASTER_STAGE_NO_BLEND | EXA_RASTER_STAGE_NO_SORT | EXA_RASTER_STAGE_SORT_ONLY)); }

int forward_render_group(const ExaRasterForwardJob* jobs, int K, int store_ctx, hipStream_t st, bool merged = false) {
    BinArgs ba[MAX_BATCH];
    RenderFwdArgs ra[MAX_BATCH];
    const ExaRasterSettings* s0 = jobs[0].settings;
    for (int k = 0; k < K; ++k) {
        const ExaRasterForwardJob& j = jobs[k];
        ba[k] = bin_args(j);
        ba[k].merged = merged ? 1 : 0;
        RenderFwdArgs& r = ra[k];
        r.grid = ba[k].grid; r.splats = ba[k].splats; r.tw = ba[k].tw; r.bw = ba[k].bw; r.capacity = j.capacity;
        r.bg = j.settings->bg; r.out_color = j.out_color; r.out_depth = j.out_depth; r.out_alpha = j.out_alpha;
        r.store_ctx = store_ctx & 1;
        r.keep_sorted_keys = j.keep_sorted_keys; r.splats2 = nullptr;
        r.src_color = r.src_depth = r.src_alpha = r.src_bg = nullptr;
    }
    int rc;
    if (stage_lists(store_ctx)) {
        // (cell_scatter_kernel also clears the zero-filled section of the bin workspace: batch owners, blended masks, touched bytes)
        EXA_TIMED(K_CELL_SCATTER, launch_cell_scatter(ba, K, st), "cell_scatter");
        if ((rc = debug_sync(s0, st, "cell_scatter"))) return rc;
        EXA_TIMED(K_SUBTILE_BIN, launch_subtile_bin(ba, K, st), "subtile_bin");
        if ((rc = debug_sync(s0, st, "subtile_bin"))) return rc;
    }
    if (stage_sort(store_ctx)) {
        EXA_TIMED(K_SORT, launch_sort_subtiles(ra, K, st), "sort_subtiles");
        if ((rc = debug_sync(s0, st, "sort_subtiles"))) return rc;
    }
    if (stage_blend(store_ctx)) {
        EXA_TIMED(K_RENDER_FWD, launch_render_fwd(ra, K, st), "render_fwd");
        if ((rc = debug_sync(s0, st, "render_fwd"))) return rc;
    }
    return 0;
}

int check_backward_job(const ExaRasterBackwardJob& j) {
    int rc = check_settings(j.settings);
    if (rc) return rc;
    rc = check_inputs(j.P, j.sh_M, j.means3D, j.shs, j.colors_precomp, j.opacities, j.scales, j.rotations,
                      j.cov3D_precomp, j.settings->sh_degree);
    if (rc) return rc;
    if (j.P == 0) return 0;
    if (!j.geom_ws || !j.tile_ws || (j.capacity > 0 && !j.bin_ws) || !j.grad_ws || !j.radii)
        return fail(EXA_RASTER_E_WORKSPACE, "workspace / radii is NULL");
    if (j.capacity % BATCH) return fail(EXA_RASTER_E_INVALID, "capacity must be a multiple of 64");
    if (!j.dL_dcolor) return fail(EXA_RASTER_E_NULLPTR, "dL_dcolor is NULL");
    if (j.grad_first < 0 || j.grad_first > j.P) return fail(EXA_RASTER_E_INVALID, "grad_first must be in 0..P");
    if (j.compose_geom_a && (j.grad_first != 0 || j.compose_P_a <= 0 || j.compose_capacity_b % BATCH))
        return fail(EXA_RASTER_E_INVALID, "composite backward: grad_first must be 0, compose_P_a > 0, compose_capacity_b a multiple of 64");
    return 0;
}

int backward_group(const ExaRasterBackwardJob* jobs, int K, int sum_shared, int dens_shared, hipStream_t st, int stage = 0) {
    RenderBwdArgs ra[MAX_BATCH];
    PreprocessBwdArgs pa[MAX_BATCH];
    const ExaRasterSettings* s0 = jobs[0].settings;
    int n = 0;
    for (int k = 0; k < K; ++k) {
        const ExaRasterBackwardJob& j = jobs[k];
        if (j.P == 0 || j.grad_first == j.P) continue;            // nothing to differentiate
        const ExaRasterSettings* s = j.settings;
        const Grid g = make_grid(s->image_width, s->image_height);
        RenderBwdArgs& r = ra[n];
        r.grid = g; r.capacity = j.capacity; r.P = j.P; r.splats = static_cast<const Splat*>(j.geom_ws);
        r.splats2 = nullptr; r.P2 = 0;
        r.tw = carve_tile_ws(const_cast<void*>(j.tile_ws), g.cells, num_chunks(j.P));
        r.bw = carve_bin_ws(const_cast<void*>(j.bin_ws), j.capacity);
        r.bg = s->bg; r.dL_dcolor = j.dL_dcolor; r.dL_ddepth = j.dL_ddepth; r.dL_dalpha = j.dL_dalpha;
        r.dL_dcolor_ind = j.dL_dcolor_indirect;
        r.used_slots = j.used_slots;
        r.partials = carve_grad_ws(j.grad_ws, j.capacity);
        r.grad_first = j.grad_first;
        if (j.compose_geom_a) {                 // composite: ids of two record arrays, A constant, B (= this job's tensors) trainable
            r.splats = static_cast<const Splat*>(j.compose_geom_a); r.P = j.compose_P_a;
            r.splats2 = static_cast<const Splat*>(j.geom_ws); r.P2 = j.P;
            r.tw = carve_tile_ws(const_cast<void*>(j.tile_ws), g.cells, 0);
            r.bw = carve_compose_ws(const_cast<void*>(j.bin_ws), j.capacity, j.compose_capacity_b);
            r.grad_first = 1;                   // selects the prefix-aware instantiation; the kernel decides on the source bit
        }
        PreprocessBwdArgs& b = pa[n];
        b.P = j.P; b.sh_M = j.sh_M; b.sh_degree = s->sh_degree; b.grid = g;
        b.tanfovx = s->tanfovx; b.tanfovy = s->tanfovy;
        b.focal_x = (float)s->image_width / (2.0f * s->tanfovx);
        b.focal_y = (float)s->image_height / (2.0f * s->tanfovy);
        b.scale_modifier = s->scale_modifier;
        b.viewmatrix = s->viewmatrix; b.projmatrix = s->projmatrix; b.campos = s->campos;
        b.means3D = j.means3D; b.shs = j.shs; b.opacities = j.opacities; b.scales = j.scales; b.rotations = j.rotations;
        b.cov3D_precomp = j.cov3D_precomp; b.radii = j.radii; b.splats = static_cast<const Splat*>(j.geom_ws);   // (composite: B's records)
        b.partials = r.partials; b.touched = r.bw.touched; b.header = r.tw.header;
        b.partial_bytes = (uint64_t)PARTIAL_BYTES * (j.compose_geom_a ? j.compose_capacity_b : j.capacity);
        b.dL_dmeans2D = j.dL_dmeans2D; b.dL_dmeans3D = j.dL_dmeans3D; b.dL_dcolors = j.dL_dcolors;
        b.dL_dopacity = j.dL_dopacity; b.dL_dscales = j.dL_dscales; b.dL_drotations = j.dL_drotations;
        b.dL_dsh = j.dL_dsh; b.dL_dcov3D = j.dL_dcov3D;
        b.dens_accum = j.densify_grad_accum; b.dens_cnt = j.densify_track_cnt; b.dens_rmax = j.densify_radius_max;
        b.accumulate = j.accumulate;
        // (a composite's grad workspace holds its partial records only: composites are never summed over views)
        b.group_scratch = j.compose_geom_a ? nullptr : reinterpret_cast<float*>(static_cast<char*>(j.grad_ws) + grad_ws_bytes(j.capacity));
        b.grad_first = j.grad_first;            // (composite: 0 -- every Gaussian of B is trainable; partials / touched / header are the composite's)
        ++n;
    }
    if (n == 0) return 0;
    int rc;
    if (!(stage & EXA_RASTER_STAGE_NO_BLEND)) {          // (as in the forward: NO_BLEND skips the blend's backward ...)
        EXA_TIMED(K_RENDER_BWD, launch_render_bwd(ra, n, st), "render_bwd");
        if ((rc = debug_sync(s0, st, "render_bwd"))) return rc;
    }
    if (!(stage & EXA_RASTER_STAGE_BLEND_ONLY)) {      // (... BLEND_ONLY = stop before the per-Gaussian chain rule)
        EXA_TIMED(K_PREPROCESS_BWD, launch_preprocess_bwd(pa, n, sum_shared, dens_shared, st), "preprocess_bwd");
        if ((rc = debug_sync(s0, st, "preprocess_bwd"))) return rc;
    }
    return 0;
}

}  // namespace

extern "C" {

int exa_raster_forward_bin_batch(const ExaRasterForwardJob* jobs, int32_t K, void* stream) {
    if (K < 0 || (K > 0 && !jobs)) return fail(EXA_RASTER_E_INVALID, "bad job list");
    for (int k = 0; k < K; ++k) {
        const int rc = check_forward_job(jobs[k], true, false);
        if (rc) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int k0 = 0; k0 < K; k0 += MAX_BATCH) {
        const int rc = forward_bin_group(jobs + k0, K - k0 < MAX_BATCH ? K - k0 : MAX_BATCH, st);
        if (rc) return rc;
    }
    return 0;
}

int exa_raster_forward_render_batch(const ExaRasterForwardJob* jobs, int32_t K, int32_t store_ctx, void* stream) {
    if (K < 0 || (K > 0 && !jobs)) return fail(EXA_RASTER_E_INVALID, "bad job list");
    for (int k = 0; k < K; ++k) {
        const int rc = check_forward_job(jobs[k], false, true);
        if (rc) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int k0 = 0; k0 < K; k0 += MAX_BATCH) {
        const int rc = forward_render_group(jobs + k0, K - k0 < MAX_BATCH ? K - k0 : MAX_BATCH, store_ctx, st);
        if (rc) return rc;
    }
    return 0;
}

int exa_raster_forward_batch(const ExaRasterForwardJob* jobs, int32_t K, int32_t store_ctx, void* stream) {
    if (K < 0 || (K > 0 && !jobs)) return fail(EXA_RASTER_E_INVALID, "bad job list");
    for (int k = 0; k < K; ++k) {
        const int rc = check_forward_job(jobs[k], true, true);
        if (rc) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int k0 = 0; k0 < K; k0 += MAX_BATCH) {
        const int n = K - k0 < MAX_BATCH ? K - k0 : MAX_BATCH;
        const bool merged = can_merge_scans(jobs + k0, n);
        int rc = stage_lists(store_ctx) ? forward_bin_group(jobs + k0, n, st, merged) : 0;
        if (rc) return rc;
        rc = forward_render_group(jobs + k0, n, store_ctx, st, merged);
        if (rc) return rc;
    }
    return 0;
}

int exa_raster_backward_batch(const ExaRasterBackwardJob* jobs, int32_t K, int32_t flags, void* stream) {
    const int32_t sum_shared = flags & 1, stage = flags & (EXA_RASTER_STAGE_NO_BLEND | EXA_RASTER_STAGE_BLEND_ONLY);
    if (K < 0 || (K > 0 && !jobs)) return fail(EXA_RASTER_E_INVALID, "bad job list");
    for (int k = 0; k < K; ++k) {
        const int rc = check_backward_job(jobs[k]);
        if (rc) return rc;
        if (sum_shared) {
            const ExaRasterBackwardJob &a = jobs[0], &b = jobs[k];
            if (b.grad_first != 0) return fail(EXA_RASTER_E_INVALID, "sum_shared and grad_first cannot be combined");
            if (b.accumulate) return fail(EXA_RASTER_E_INVALID, "sum_shared and accumulate cannot be combined");
            if (b.compose_geom_a) return fail(EXA_RASTER_E_INVALID, "sum_shared and composite jobs cannot be combined");
            if (a.P != b.P || a.sh_M != b.sh_M || a.means3D != b.means3D || a.shs != b.shs || a.opacities != b.opacities ||
                a.colors_precomp != b.colors_precomp || a.scales != b.scales || a.rotations != b.rotations ||
                a.cov3D_precomp != b.cov3D_precomp || a.settings->sh_degree != b.settings->sh_degree ||
                a.settings->scale_modifier != b.settings->scale_modifier)
                return fail(EXA_RASTER_E_INVALID, "sum_shared needs K views of the same Gaussian tensors");
        }
    }
    if (sum_shared && K > MAX_BATCH) return fail(EXA_RASTER_E_INVALID, "sum_shared supports at most 8 views per call");
    // Densification statistics are plain read-modify-writes per job: two jobs of one call must not update the same
    // array (their workgroups run concurrently).  The one supported form of sharing: sum_shared with ALL K views
    // accumulating into the same three arrays (K views of one model -> one set of statistics), summed inside the kernel.
    int dens_shared = 0;
    {
        int n_dens = 0, n_same = 0;
        for (int k = 0; k < K; ++k) {
            const ExaRasterBackwardJob& b = jobs[k];
            if (!b.densify_grad_accum && !b.densify_track_cnt && !b.densify_radius_max) continue;
            ++n_dens;
            if (b.densify_grad_accum == jobs[0].densify_grad_accum && b.densify_track_cnt == jobs[0].densify_track_cnt &&
                b.densify_radius_max == jobs[0].densify_radius_max) ++n_same;
        }
        if (sum_shared && K > 1 && n_dens == K && n_same == K) dens_shared = 1;
        else if (n_dens > 1) {
            for (int i = 0; i < K; ++i)
                for (int j = i + 1; j < K; ++j) {
                    const float* pi[3] = {jobs[i].densify_grad_accum, jobs[i].densify_track_cnt, jobs[i].densify_radius_max};
                    const float* pj[3] = {jobs[j].densify_grad_accum, jobs[j].densify_track_cnt, jobs[j].densify_radius_max};
                    for (int u = 0; u < 3; ++u)
                        for (int v = 0; v < 3; ++v)
                            if (pi[u] && pi[u] == pj[v])
                                return fail(EXA_RASTER_E_ALIAS, "two jobs of one batch update the same densification-statistics "
                                            "array (supported only as sum_shared with the same three arrays in every job)");
                }
        }
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int k0 = 0; k0 < K; k0 += MAX_BATCH) {
        const int rc = backward_group(jobs + k0, K - k0 < MAX_BATCH ? K - k0 : MAX_BATCH, sum_shared, dens_shared, st, stage);
        if (rc) return rc;
    }
    return 0;
}

// ---- single-render entry points: a batch of one ------------------------------------------------------------
static ExaRasterForwardJob one_job(const ExaRasterSettings* s, int32_t P, int32_t sh_M, const float* means3D,
                                   const float* shs, const float* colors_precomp, const float* opacities,
                                   const float* scales, const float* rotations, const float* cov3D_precomp,
                                   int32_t* radii, void* geom_ws, void* tile_ws, void* bin_ws, uint64_t capacity,
                                   float* out_color, float* out_depth, float* out_alpha) {
    ExaRasterForwardJob j{};                 // optional fields (is_vis, host_header ...) off
    j.settings = s; j.P = P; j.sh_M = sh_M; j.means3D = means3D; j.shs = shs; j.colors_precomp = colors_precomp;
    j.opacities = opacities; j.scales = scales; j.rotations = rotations; j.cov3D_precomp = cov3D_precomp;
    j.radii = radii; j.geom_ws = geom_ws; j.tile_ws = tile_ws; j.bin_ws = bin_ws; j.capacity = capacity;
    j.out_color = out_color; j.out_depth = out_depth; j.out_alpha = out_alpha;
    j.host_header = nullptr; j.header_tag = 0u; j.keep_sorted_keys = 0;
    return j;
}

int exa_raster_forward_bin(const ExaRasterSettings* s, int32_t P, int32_t sh_M, const float* means3D,
                           const float* shs, const float* colors_precomp, const float* opacities,
                           const float* scales, const float* rotations, const float* cov3D_precomp,
                           int32_t* radii, void* geom_ws, void* tile_ws, void* stream) {
    const ExaRasterForwardJob j = one_job(s, P, sh_M, means3D, shs, colors_precomp, opacities, scales, rotations,
                                          cov3D_precomp, radii, geom_ws, tile_ws, nullptr, 0, nullptr, nullptr, nullptr);
    return exa_raster_forward_bin_batch(&j, 1, stream);
}

int exa_raster_forward_render(const ExaRasterSettings* s, int32_t P, const void* geom_ws, void* tile_ws, void* bin_ws,
                              uint64_t capacity, float* out_color, float* out_depth, float* out_alpha,
                              int32_t store_ctx, void* stream) {
    const ExaRasterForwardJob j = one_job(s, P, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                          const_cast<void*>(geom_ws), tile_ws, bin_ws, capacity, out_color, out_depth,
                                          out_alpha);
    return exa_raster_forward_render_batch(&j, 1, store_ctx, stream);
}

int exa_raster_forward(const ExaRasterSettings* s, int32_t P, int32_t sh_M, const float* means3D, const float* shs,
                       const float* colors_precomp, const float* opacities, const float* scales,
                       const float* rotations, const float* cov3D_precomp, int32_t* radii, void* geom_ws,
                       void* tile_ws, void* bin_ws, uint64_t capacity, float* out_color,
                       float* out_depth, float* out_alpha, int32_t store_ctx, void* stream) {
    const ExaRasterForwardJob j = one_job(s, P, sh_M, means3D, shs, colors_precomp, opacities, scales, rotations,
                                          cov3D_precomp, radii, geom_ws, tile_ws, bin_ws, capacity, out_color, out_depth,
                                          out_alpha);
    return exa_raster_forward_batch(&j, 1, store_ctx, stream);
}

int exa_raster_backward(const ExaRasterSettings* s, int32_t P, int32_t sh_M, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* opacities, const float* scales,
                        const float* rotations, const float* cov3D_precomp, const int32_t* radii,
                        const void* geom_ws, const void* tile_ws, const void* bin_ws, uint64_t capacity,
                        const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                        void* grad_ws, float* dL_dmeans2D, float* dL_dmeans3D, float* dL_dcolors, float* dL_dopacity,
                        float* dL_dscales, float* dL_drotations, float* dL_dsh, float* dL_dcov3D, void* stream) {
    ExaRasterBackwardJob j{};                // optional fields (dL_dcolor_indirect, accumulate ...) off
    j.settings = s; j.P = P; j.sh_M = sh_M; j.means3D = means3D; j.shs = shs; j.colors_precomp = colors_precomp;
    j.opacities = opacities; j.scales = scales; j.rotations = rotations; j.cov3D_precomp = cov3D_precomp;
    j.radii = radii; j.geom_ws = geom_ws; j.tile_ws = tile_ws; j.bin_ws = bin_ws; j.capacity = capacity;
    j.dL_dcolor = dL_dcolor; j.dL_ddepth = dL_ddepth; j.dL_dalpha = dL_dalpha; j.grad_ws = grad_ws;
    j.dL_dmeans2D = dL_dmeans2D; j.dL_dmeans3D = dL_dmeans3D; j.dL_dcolors = dL_dcolors; j.dL_dopacity = dL_dopacity;
    j.dL_dscales = dL_dscales; j.dL_drotations = dL_drotations; j.dL_dsh = dL_dsh; j.dL_dcov3D = dL_dcov3D;
    j.densify_grad_accum = nullptr; j.densify_track_cnt = nullptr; j.densify_radius_max = nullptr;
    j.grad_first = 0;
    j.compose_geom_a = nullptr; j.compose_P_a = 0; j.compose_capacity_b = 0;
    return exa_raster_backward_batch(&j, 1, 0, stream);
}

int exa_raster_compose_sizes(int32_t W, int32_t H, uint64_t capacity, uint64_t capacity_b, ExaRasterWorkspaceSizes* out) {
    if (!out) return fail(EXA_RASTER_E_NULLPTR, "out is NULL");
    if (W < 0 || H < 0) return fail(EXA_RASTER_E_INVALID, "negative size");
    if (capacity % BATCH || capacity_b % BATCH) return fail(EXA_RASTER_E_INVALID, "capacity must be a multiple of 64");
    out->geom_bytes = 0;
    out->tile_bytes = tile_ws_bytes(make_grid(W, H).cells, 0);
    out->bin_bytes = compose_bin_bytes(capacity, capacity_b);
    out->grad_bytes = grad_ws_bytes(capacity_b);
    return 0;
}

int exa_raster_forward_compose_batch(const ExaRasterComposeJob* jobs, int32_t K, int32_t store_ctx, void* stream) {
    if (K < 0 || (K > 0 && !jobs)) return fail(EXA_RASTER_E_INVALID, "bad job list");
    for (int k = 0; k < K; ++k) {
        const ExaRasterComposeJob& j = jobs[k];
        const int rc = check_settings(j.settings);
        if (rc) return rc;
        if (j.P_a <= 0 || j.P_b <= 0) return fail(EXA_RASTER_E_INVALID, "composite: both sources need Gaussians");
        if (!j.geom_a || !j.tile_a || !j.bin_a || !j.geom_b || !j.tile_b || !j.bin_b || !j.tile_ws || !j.bin_ws)
            return fail(EXA_RASTER_E_WORKSPACE, "composite: workspace is NULL");
        if (j.capacity % BATCH || j.capacity_a % BATCH || j.capacity_b % BATCH)
            return fail(EXA_RASTER_E_INVALID, "capacity must be a multiple of 64");
        if (!j.out_color || !j.out_depth || !j.out_alpha) return fail(EXA_RASTER_E_NULLPTR, "output image is NULL");
        const int n_src = (j.a_color != nullptr) + (j.a_depth != nullptr) + (j.a_alpha != nullptr) + (j.a_bg != nullptr);
        if (n_src != 0 && n_src != 4) return fail(EXA_RASTER_E_INVALID, "composite: pass all of a_color / a_depth / a_alpha / a_bg or none");
        if (j.a_color == j.out_color && j.a_color) return fail(EXA_RASTER_E_ALIAS, "composite: a_color aliases out_color");
        if ((j.radii_out && (!j.radii_a || !j.radii_b)) || (j.is_vis_out && (!j.is_vis_a || !j.is_vis_b)))
            return fail(EXA_RASTER_E_NULLPTR, "composite: radii_out / is_vis_out need the sources' arrays");
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int k0 = 0; k0 < K; k0 += MAX_BATCH) {
        const int n = K - k0 < MAX_BATCH ? K - k0 : MAX_BATCH;
        ComposeArgs ca[MAX_BATCH];
        RenderFwdArgs ra[MAX_BATCH];
        for (int k = 0; k < n; ++k) {
            const ExaRasterComposeJob& j = jobs[k0 + k];
            const Grid g = make_grid(j.settings->image_width, j.settings->image_height);
            ComposeArgs& c = ca[k];
            c.grid = g;
            c.tw_a = carve_tile_ws(const_cast<void*>(j.tile_a), g.cells, num_chunks(j.P_a));
            c.tw_b = carve_tile_ws(const_cast<void*>(j.tile_b), g.cells, num_chunks(j.P_b));
            c.bw_a = carve_bin_ws(const_cast<void*>(j.bin_a), j.capacity_a);
            c.bw_b = carve_bin_ws(const_cast<void*>(j.bin_b), j.capacity_b);
            c.tw = carve_tile_ws(j.tile_ws, g.cells, 0);
            c.bw = carve_compose_ws(j.bin_ws, j.capacity, j.capacity_b);
            c.capacity = j.capacity; c.capacity_b = j.capacity_b;
            c.host_hdr = static_cast<uint32_t*>(j.host_header); c.hdr_tag = j.header_tag;
            c.src_color = j.a_color; c.src_bg = j.a_bg; c.bg = j.settings->bg;
            c.P_a = j.P_a; c.P_b = j.P_b;
            c.radii_a = j.radii_a; c.radii_b = j.radii_b; c.radii_out = j.radii_out;
            c.vis_a = j.is_vis_a; c.vis_b = j.is_vis_b; c.vis_out = j.is_vis_out;
            RenderFwdArgs& r = ra[k];
            r.grid = g; r.splats = static_cast<const Splat*>(j.geom_a); r.splats2 = static_cast<const Splat*>(j.geom_b);
            r.tw = c.tw; r.bw = c.bw; r.capacity = j.capacity;
            r.bg = j.settings->bg; r.out_color = j.out_color; r.out_depth = j.out_depth; r.out_alpha = j.out_alpha;
            r.store_ctx = store_ctx & 1; r.keep_sorted_keys = 0;
            r.src_color = j.a_color; r.src_depth = j.a_depth; r.src_alpha = j.a_alpha; r.src_bg = j.a_bg;
        }
        int rc;
        const int parts = (stage_lists(store_ctx) ? 1 : 0) | (stage_sort(store_ctx) ? 2 : 0);
        if (parts) {
            EXA_TIMED(K_CELL_SCATTER, launch_compose(ca, n, st, parts), "compose");
            if ((rc = debug_sync(jobs[k0].settings, st, "compose"))) return rc;
        }
        if (stage_blend(store_ctx)) {
            EXA_TIMED(K_RENDER_FWD, launch_render_fwd(ra, n, st), "render_fwd (composite)");
            if ((rc = debug_sync(jobs[k0].settings, st, "render_fwd (composite)"))) return rc;
        }
    }
    return 0;
}

int exa_raster_read_header_async(const void* tile_ws, void* host_dst16, void* stream) {
    if (!tile_ws || !host_dst16) return fail(EXA_RASTER_E_NULLPTR, "read_header_async: NULL pointer");
    EXA_HIP(hipMemcpyAsync(host_dst16, tile_ws, 16, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)),
            "read_header_async");
    return 0;
}

int exa_raster_header_status(const ExaRasterHeader* h) {
    if (!h) return fail(EXA_RASTER_E_NULLPTR, "header is NULL");
    if (h->overflow) {
        snprintf(g_err, sizeof(g_err), "exa_raster: instance-buffer overflow (the call needs capacity %u)", h->num_rendered);
        return EXA_RASTER_E_OVERFLOW;
    }
    return 0;
}

int exa_raster_camera_block(const float* R, const float* t, const float* proj16_host, float* viewmatrix_out,
                            float* projmatrix_out, float* campos_out, const float* focal, float fx_expected,
                            float fy_expected, void* host_flag, uint32_t flag_tag, void* stream) {
    if (!R || !t || !proj16_host || !viewmatrix_out || !projmatrix_out || !campos_out)
        return fail(EXA_RASTER_E_NULLPTR, "camera_block: NULL pointer");
    if ((focal != nullptr) != (host_flag != nullptr))
        return fail(EXA_RASTER_E_INVALID, "camera_block: pass both focal and host_flag, or neither");
    Proj16 p;
    memcpy(p.m, proj16_host, sizeof(p.m));
    EXA_HIP(launch_camera_block(R, t, p, viewmatrix_out, projmatrix_out, campos_out, focal, fx_expected, fy_expected,
                                static_cast<uint32_t*>(host_flag), flag_tag, static_cast<hipStream_t>(stream)),
            "camera_block");
    return 0;
}

namespace {
struct Ptr16 { const void* p[16]; };
__global__ void store_pointers_kernel(const void** table, Ptr16 v, int n) {
    if ((int)threadIdx.x < n) table[threadIdx.x] = v.p[threadIdx.x];
}
// dst = table[*counter mod n_rows]; *counter = (that row + 1) mod n_rows.  One wave: every lane has read the counter before lane 0
// writes it (program order of a wave).
__global__ __launch_bounds__(64) void select_row_kernel(const float* __restrict__ table, int n_rows, int row_floats,
                                                        int32_t* __restrict__ counter, float* __restrict__ dst) {
    const int r = (int)((uint32_t)*counter % (uint32_t)n_rows);
    for (int i = threadIdx.x; i < row_floats; i += 64) dst[i] = table[(size_t)r * row_floats + i];
    if (threadIdx.x == 0) *counter = r + 1 == n_rows ? 0 : r + 1;
}
}  // namespace

int exa_raster_select_row(const float* table, int32_t n_rows, int32_t row_floats, int32_t* counter, float* dst, void* stream) {
    if (!table || !counter || !dst) return fail(EXA_RASTER_E_NULLPTR, "select_row: NULL pointer");
    if (n_rows <= 0 || row_floats <= 0) return fail(EXA_RASTER_E_INVALID, "select_row: n_rows and row_floats must be positive");
    select_row_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(table, n_rows, row_floats, counter, dst);
    EXA_HIP(hipGetLastError(), "select_row");
    return 0;
}

int exa_raster_store_pointers(void* table, const void* const* ptrs, int32_t n, void* stream) {
    if (!table || !ptrs) return fail(EXA_RASTER_E_NULLPTR, "store_pointers: NULL pointer");
    if (n < 0 || n > 16) return fail(EXA_RASTER_E_INVALID, "store_pointers: n must be 0..16");
    if (n == 0) return 0;
    Ptr16 v;
    for (int i = 0; i < 16; ++i) v.p[i] = i < n ? ptrs[i] : nullptr;
    store_pointers_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(static_cast<const void**>(table), v, n);
    EXA_HIP(hipGetLastError(), "store_pointers");
    return 0;
}

int exa_raster_host_device_pointer(void* host_ptr, void** device_ptr_out) {
    if (!host_ptr || !device_ptr_out) return fail(EXA_RASTER_E_NULLPTR, "host_device_pointer: NULL pointer");
    EXA_HIP(hipHostGetDevicePointer(device_ptr_out, host_ptr, 0), "hipHostGetDevicePointer");
    return 0;
}

int exa_raster_read_header_full_async(const void* tile_ws, void* host_dst32, void* stream) {
    if (!tile_ws || !host_dst32) return fail(EXA_RASTER_E_NULLPTR, "read_header_full_async: NULL pointer");
    EXA_HIP(hipMemcpyAsync(host_dst32, tile_ws, sizeof(ExaRasterHeader), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)),
            "read_header_full_async");
    return 0;
}

int exa_raster_mark_visible(const ExaRasterSettings* s, int32_t P, const float* means3D, uint8_t* present,
                            void* stream) {
    int rc = check_settings(s);
    if (rc) return rc;
    if (P < 0) return fail(EXA_RASTER_E_INVALID, "P < 0");
    if (P > 0 && (!means3D || !present)) return fail(EXA_RASTER_E_NULLPTR, "means3D / present is NULL");
    hipStream_t st = static_cast<hipStream_t>(stream);
    EXA_HIP(launch_mark_visible(P, means3D, s->viewmatrix, present, st), "mark_visible");
    return debug_sync(s, st, "mark_visible");
}

int exa_raster_densify_stats(int32_t P, const float* dL_dmeans2D, const int32_t* radii, float* xyz_grad_accum,
                             float* track_cnt, float* radius_max, void* stream) {
    if (P < 0) return fail(EXA_RASTER_E_INVALID, "P < 0");
    if (P > 0 && !radii) return fail(EXA_RASTER_E_NULLPTR, "radii is NULL");
    if (P > 0 && xyz_grad_accum && !dL_dmeans2D) return fail(EXA_RASTER_E_NULLPTR, "dL_dmeans2D is NULL");
    EXA_HIP(launch_densify_stats(P, dL_dmeans2D, radii, xyz_grad_accum, track_cnt, radius_max,
                                 static_cast<hipStream_t>(stream)), "densify_stats");
    return 0;
}

int exa_ssim_forward(int32_t N, int32_t H, int32_t W, const float* img1, const float* img2, float* ssim_map,
                     float* dm_dmu1, float* dm_dE11, float* dm_dE12, void* stream) {
    if (N < 0 || H < 0 || W < 0) return fail(EXA_RASTER_E_INVALID, "negative size");
    if ((int64_t)N * H * W > 0 && (!img1 || !img2 || !ssim_map)) return fail(EXA_RASTER_E_NULLPTR, "img / map is NULL");
    if ((dm_dmu1 != nullptr) != (dm_dE11 != nullptr) || (dm_dmu1 != nullptr) != (dm_dE12 != nullptr))
        return fail(EXA_RASTER_E_INVALID, "pass all three derivative maps or none");
    EXA_HIP(launch_ssim_fwd(N, H, W, img1, img2, ssim_map, dm_dmu1, dm_dE11, dm_dE12, static_cast<hipStream_t>(stream)),
            "ssim_fwd");
    return 0;
}

int exa_ssim_backward(int32_t N, int32_t H, int32_t W, const float* img1, const float* img2, const float* dL_dmap,
                      const float* dm_dmu1, const float* dm_dE11, const float* dm_dE12, float* dL_dimg1,
                      void* stream) {
    if (N < 0 || H < 0 || W < 0) return fail(EXA_RASTER_E_INVALID, "negative size");
    if ((int64_t)N * H * W > 0 && (!img1 || !img2 || !dL_dmap || !dm_dmu1 || !dm_dE11 || !dm_dE12 || !dL_dimg1))
        return fail(EXA_RASTER_E_NULLPTR, "ssim backward: NULL argument");
    EXA_HIP(launch_ssim_bwd(N, H, W, img1, img2, dL_dmap, dm_dmu1, dm_dE11, dm_dE12, dL_dimg1,
                            static_cast<hipStream_t>(stream)), "ssim_bwd");
    return 0;
}

static int check_crop(int32_t B, int32_t C, int32_t H, int32_t W, const int32_t* crop) {
    if (B < 0 || C < 0 || H < 0 || W < 0) return fail(EXA_RASTER_E_INVALID, "negative size");
    if (!crop) return fail(EXA_RASTER_E_NULLPTR, "crop is NULL");
    if (crop[0] < 0 || crop[1] < 0 || crop[2] < 0 || crop[3] < 0 || crop[0] + crop[2] > W || crop[1] + crop[3] > H)
        return fail(EXA_RASTER_E_INVALID, "crop window outside the image");
    return 0;
}

int64_t exa_photo_loss_blocks(int32_t B, int32_t C, int32_t crop_w, int32_t crop_h) {
    if (B < 0 || C < 0 || crop_w < 0 || crop_h < 0) return 0;
    return (int64_t)B * C * ((crop_w + 31) / 32) * ((crop_h + 31) / 32);
}

int exa_photo_loss_forward(int32_t B, int32_t C, int32_t H, int32_t W, const int32_t* crop, const float* img_out,
                           const float* img_target, const float* l1_weight, const float* ssim_mask, float* maps_ws,
                           float* partials, void* stream) {
    int rc = check_crop(B, C, H, W, crop);
    if (rc) return rc;
    if ((int64_t)B * C * crop[2] * crop[3] > 0 && (!img_out || !img_target || !maps_ws || !partials))
        return fail(EXA_RASTER_E_NULLPTR, "photo loss: NULL argument");
    EXA_HIP(launch_photo_loss(B, C, H, W, crop, img_out, img_target, l1_weight, ssim_mask, 0.f, 0.f, maps_ws, partials,
                              nullptr, 0, static_cast<hipStream_t>(stream)), "photo_stats");
    return 0;
}

int exa_photo_loss_grad(int32_t B, int32_t C, int32_t H, int32_t W, const int32_t* crop, const float* img_out,
                        const float* img_target, const float* l1_weight, const float* ssim_mask, float w_l1, float w_ssim,
                        const float* maps_ws, float* dL_dimg, void* stream) {
    int rc = check_crop(B, C, H, W, crop);
    if (rc) return rc;
    if ((int64_t)B * C * crop[2] * crop[3] > 0 && (!img_out || !img_target || !maps_ws || !dL_dimg))
        return fail(EXA_RASTER_E_NULLPTR, "photo loss: NULL argument");
    EXA_HIP(launch_photo_loss(B, C, H, W, crop, img_out, img_target, l1_weight, ssim_mask, w_l1, w_ssim,
                              const_cast<float*>(maps_ws), nullptr, dL_dimg, 1, static_cast<hipStream_t>(stream)), "photo_grad");
    return 0;
}

int exa_l1_forward(int32_t B, int32_t C, int32_t H, int32_t W, const int32_t* crop, const float* img_out,
                   const float* img_target, const float* mask, const float* bg, float* l1_map, void* stream) {
    int rc = check_crop(B, C, H, W, crop);
    if (rc) return rc;
    if ((int64_t)B * C * crop[2] * crop[3] > 0 && (!img_out || !img_target || !l1_map))
        return fail(EXA_RASTER_E_NULLPTR, "l1: NULL argument");
    EXA_HIP(launch_l1(B, C, H, W, crop, img_out, img_target, mask, bg, nullptr, l1_map, 0, static_cast<hipStream_t>(stream)),
            "l1_fwd");
    return 0;
}

int exa_l1_backward(int32_t B, int32_t C, int32_t H, int32_t W, const int32_t* crop, const float* img_out,
                    const float* img_target, const float* mask, const float* bg, const float* dL_dmap, float* dL_dimg,
                    void* stream) {
    int rc = check_crop(B, C, H, W, crop);
    if (rc) return rc;
    if ((int64_t)B * C * crop[2] * crop[3] > 0 && (!img_out || !img_target || !dL_dmap || !dL_dimg))
        return fail(EXA_RASTER_E_NULLPTR, "l1: NULL argument");
    EXA_HIP(launch_l1(B, C, H, W, crop, img_out, img_target, mask, bg, dL_dmap, dL_dimg, 1, static_cast<hipStream_t>(stream)),
            "l1_bwd");
    return 0;
}

int exa_raster_timing_enable(int32_t on) {
    if (on && !g_t.created) {
        for (int i = 0; i < K_COUNT; ++i)
            for (int j = 0; j < 2; ++j) EXA_HIP(hipEventCreate(&g_t.ev[i][j]), "hipEventCreate");
        g_t.created = true;
    }
    g_t.enabled = on != 0;
    for (int i = 0; i < K_COUNT; ++i) g_t.used[i] = false;
    return 0;
}

int exa_raster_timing_read(float* ms_out, int32_t n) {
    if (!ms_out) return fail(EXA_RASTER_E_NULLPTR, "ms_out is NULL");
    for (int i = 0; i < n; ++i) {
        ms_out[i] = -1.0f;
        if (i < K_COUNT && g_t.created && g_t.used[i]) {
            EXA_HIP(hipEventSynchronize(g_t.ev[i][1]), "hipEventSynchronize");
            EXA_HIP(hipEventElapsedTime(&ms_out[i], g_t.ev[i][0], g_t.ev[i][1]), "hipEventElapsedTime");
            g_t.used[i] = false;
        }
    }
    return 0;
}

const char* exa_raster_timing_name(int32_t slot) { return (slot >= 0 && slot < K_COUNT) ? k_names[slot] : ""; }

}  // extern "C"
