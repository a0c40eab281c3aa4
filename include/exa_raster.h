/*
 * exa_raster.h -- C ABI of the MI355X-native differentiable 3D-Gaussian rasterizer.
 *
 * This is the drop-in boundary for the one native component on ExAvatar's render path: the
 * third-party CUDA extension `diff_gaussian_rasterization_depth` that the reference imports at
 * avatar/common/nets/module.py:11 and calls at module.py:623-640 (forward) and through autograd
 * from avatar/main/train.py:46 (backward).  Upstream binds three C++ entry points through pybind
 * (`rasterize_gaussians`, `rasterize_gaussians_backward`, `mark_visible`); the functions below are
 * what a ctypes / cffi / pybind stub binds instead.
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, a `hipStream_t` passed as `void*`.
 *   - every pointer marked [dev] is a device pointer owned by the caller (PyTorch's caching
 *     allocator in the Python binding); the library allocates nothing and keeps no state between
 *     calls, so any number of forward contexts may be live before their backward runs
 *     (ExAvatar keeps five, avatar/main/model.py:130-162).
 *   - all floating point is fp32, indices int32/uint32, contiguous row-major tensors with the
 *     shapes the reference passes (module.py:632-640): means3D[P,3], opacities[P,1], scales[P,3],
 *     rotations[P,4] (w,x,y,z), colors_precomp[P,3], shs[P,M,3], cov3D_precomp[P,6].
 *   - work is enqueued on `stream`; no call synchronises the device unless `settings->debug != 0`.
 *   - return value: 0 = ok; < 0 = invalid argument (EXA_RASTER_E_*); > 0 = HIP error code (hipError_t).
 *     Instance-buffer overflow is latched in the device-side header (see exa_raster_forward_render): the launch
 *     calls never synchronise, so they cannot return it; exa_raster_header_status() turns a header that the
 *     caller copied to the host (exa_raster_read_header_async) into EXA_RASTER_E_OVERFLOW.
 *   - `settings->prefiltered` is accepted and IGNORED: upstream uses it only to assert that the caller already
 *     removed the Gaussians behind the near plane; this library culls them itself in every call (the reference
 *     always passes False, module.py:620).
 *   - `settings->scale_modifier`: dL_dscales is the derivative with respect to `scales` itself (chain rule through
 *     mod * scale).  Upstream returns the derivative with respect to (mod * scale), i.e. it omits the factor mod;
 *     the two agree for the reference, which passes 1.0 (module.py:615).  A caller that needs upstream's quirk
 *     divides dL_dscales by scale_modifier.
 */
#ifndef EXA_RASTER_H
#define EXA_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXA_RASTER_VERSION 139          /* 0.1.3.9: tile_bytes of exa_raster_workspace_sizes grew (launch-order cursors per XCD region, one class code per sub-tile); no signature changed; 0.1.3.8: grad_bytes of exa_raster_workspace_sizes includes the 92 B x P group scratch of summed batches of more than four views; 0.1.3.7: ExaRasterComposeJob.radii_out / is_vis_out; 0.1.3.6: exa_raster_select_row; 0.1.3.5: EXA_RASTER_STAGE_* bits of store_ctx; 0.1.3.4: ExaRasterBackwardJob.used_slots; 0.1.3.3: ExaRasterForwardJob.is_vis, ExaRasterBackwardJob.accumulate; 0.1.3.2: ExaRasterBackwardJob.dL_dcolor_indirect, exa_raster_store_pointers, ExaRasterComposeJob.a_color .. a_bg; 0.1.3.1: composite renders (exa_raster_forward_compose_batch); 0.1.3: header.num_tile_instances, EXA_RASTER_E_OVERFLOW / _E_ALIAS, exa_raster_camera_block,
                                           exa_raster_header_status; 0.1.2: ExaRasterBackwardJob.grad_first; .1: exa_raster_read_header_async */
#define EXA_RASTER_TILE 16              /* 16x16 pixel tiles (upstream BLOCK_X/BLOCK_Y) */

#define EXA_RASTER_E_INVALID (-1)
#define EXA_RASTER_E_NULLPTR (-2)
#define EXA_RASTER_E_WORKSPACE (-3)
#define EXA_RASTER_E_OVERFLOW (-4)      /* instance-buffer overflow: only ever returned by exa_raster_header_status(), which
                                           interprets a header that was copied to the host; the launch calls themselves
                                           never return it (they do not synchronise) */
#define EXA_RASTER_E_ALIAS (-5)         /* two jobs of one batched backward call update the same densification-statistics
                                           arrays (the update is a plain read-modify-write per job) */

/* Mirrors the 12-field `GaussianRasterizationSettings` NamedTuple built at module.py:609-622.
 * The four tensor-valued fields stay on the device (the reference builds them with torch ops on
 * the GPU, module.py:604-608), so no host read-back is needed to fill this struct. */
typedef struct ExaRasterSettings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    const float* bg;          /* [dev] float[3] */
    float scale_modifier;
    const float* viewmatrix;  /* [dev] float[16], row-major of the [4,4] tensor (= world->camera, transposed) */
    const float* projmatrix;  /* [dev] float[16], row-major of view^T @ proj^T */
    int32_t sh_degree;
    const float* campos;      /* [dev] float[3] */
    int32_t prefiltered;
    int32_t debug;            /* != 0: hipStreamSynchronize + error check after every kernel */
} ExaRasterSettings;

/* Byte sizes of the caller-allocated workspaces. */
typedef struct ExaRasterWorkspaceSizes {
    uint64_t geom_bytes;   /* per-Gaussian splat records, 64 B * P                         */
    uint64_t tile_bytes;   /* header, (chunk, cell) count matrix, prefixes, per-sub-tile ranges   */
    uint64_t bin_bytes;    /* keys, sorted ids, cell buckets, batch owners / masks, checkpoints: ~50 B * capacity */
    uint64_t grad_bytes;   /* backward scratch: per-instance partial sums, 40 B * capacity, + 92 B * P (sum of the second
                              group of views of exa_raster_backward_batch(sum_shared = 1) with more than four views) */
} ExaRasterWorkspaceSizes;

/* Device-side header at the start of the tile workspace (readable with a 28-byte D2H copy). */
typedef struct ExaRasterHeader {
    uint32_t num_rendered;   /* instance capacity this call needs: 64 * batch slots (role of upstream's
                                num_rendered: what the caller sizes the bin workspace with)   */
    uint32_t overflow;       /* != 0: num_rendered exceeded the capacity, outputs of this call invalid */
    uint32_t max_tile_list;  /* number of (Gaussian, 64x64 cell) entries                     */
    uint32_t num_visible;    /* V = Gaussians with radius > 0                                */
    uint32_t num_instances;  /* D = (Gaussian, 8x8 sub-tile) instances actually emitted      */
    uint32_t active_cells;   /* 64x64-pixel cells that hold at least one instance            */
    uint32_t num_tile_instances; /* sum over visible Gaussians of the 16x16 tiles of their rect: upstream's
                                num_rendered, the D of the byte model in SURVEY.md 8(d)          */
} ExaRasterHeader;

/* 0 if the header copy `h` (host memory) reports a complete render, EXA_RASTER_E_OVERFLOW if the call needed
 * h->num_rendered instances but was given fewer (its outputs are invalid; re-run it with that capacity). */
int exa_raster_header_status(const ExaRasterHeader* h);

int exa_raster_version(void);

/* Thread-local description of the last non-zero status returned on this thread. */
const char* exa_raster_last_error(void);

/* Sizes for P Gaussians, a W x H image and room for `capacity` tile instances. */
int exa_raster_workspace_sizes(int32_t P, int32_t W, int32_t H, uint64_t capacity,
                               ExaRasterWorkspaceSizes* out);

/*
 * Forward, stage 1 (replaces upstream preprocessCUDA + InclusiveSum): per-Gaussian cull, EWA
 * projection, radius, tile rect, optional SH colour; per-cell instance counts and their prefix.
 * After it completes, the header in `tile_ws` holds num_rendered so the caller can size `bin_ws`
 * (upstream reads the same number back to the host at this point).
 * Exactly one of shs / colors_precomp and one of (scales+rotations) / cov3D_precomp is non-NULL.
 */
int exa_raster_forward_bin(const ExaRasterSettings* settings, int32_t P, int32_t sh_M,
                           const float* means3D, const float* shs, const float* colors_precomp,
                           const float* opacities, const float* scales, const float* rotations,
                           const float* cov3D_precomp,
                           int32_t* radii,          /* [dev] out int32[P] */
                           void* geom_ws, void* tile_ws, void* stream);

/*
 * Forward, stage 2 (replaces duplicateWithKeys + SortPairs + identifyTileRanges + renderCUDA):
 * scatter Gaussians into cell buckets and then sub-tile buckets, depth-sort every bucket in LDS,
 * blend front to back.  (geom_ws is logically const; the call fills one reserved field per record.)
 * If the header's num_rendered > capacity nothing is rendered and header.overflow is set.
 * `store_ctx` != 0 additionally writes what the backward pass needs (per-batch pixel checkpoints,
 * batches entered per sub-tile); pass 0 for inference.
 */
int exa_raster_forward_render(const ExaRasterSettings* settings, int32_t P,
                              const void* geom_ws, void* tile_ws, void* bin_ws, uint64_t capacity,
                              float* out_color,   /* [dev] float[3,H,W] */
                              float* out_depth,   /* [dev] float[1,H,W] */
                              float* out_alpha,   /* [dev] float[1,H,W] */
                              int32_t store_ctx, void* stream);

/* Stage 1 + stage 2 back to back with a fixed capacity: no host round trip, hipGraph-capturable. */
int exa_raster_forward(const ExaRasterSettings* settings, int32_t P, int32_t sh_M,
                       const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, const float* rotations,
                       const float* cov3D_precomp, int32_t* radii,
                       void* geom_ws, void* tile_ws, void* bin_ws, uint64_t capacity,
                       float* out_color, float* out_depth, float* out_alpha,
                       int32_t store_ctx, void* stream);

/*
 * Backward (replaces upstream BACKWARD::render + BACKWARD::preprocess).  Consumes the workspaces
 * a forward call with store_ctx != 0 filled.  dL_ddepth / dL_dalpha may be NULL (= zeros; the
 * ExAvatar training case).  Gradient outputs that do not apply
 * (dL_dsh without shs, dL_dcov3D without cov3D_precomp, ...) may be NULL.  All non-NULL outputs
 * are fully written (zeros for culled Gaussians).  If the forward call overflowed its instance capacity
 * (header.overflow != 0) every gradient of that render is written as zero.
 * dL_dmeans2D[P,3]: (x, y) = dL/dpix * (W/2, H/2), z = 0 -- the densification signal the reference
 * reads at avatar/main/train.py:51.
 */
int exa_raster_backward(const ExaRasterSettings* settings, int32_t P, int32_t sh_M,
                        const float* means3D, const float* shs, const float* colors_precomp,
                        const float* opacities, const float* scales, const float* rotations,
                        const float* cov3D_precomp, const int32_t* radii,
                        const void* geom_ws, const void* tile_ws, const void* bin_ws,
                        uint64_t capacity,        /* the capacity bin_ws was carved with in forward */
                        const float* dL_dcolor,   /* [dev] float[3,H,W] */
                        const float* dL_ddepth,   /* [dev] float[1,H,W] or NULL */
                        const float* dL_dalpha,   /* [dev] float[1,H,W] or NULL */
                        void* grad_ws,
                        float* dL_dmeans2D, float* dL_dmeans3D, float* dL_dcolors, float* dL_dopacity,
                        float* dL_dscales, float* dL_drotations, float* dL_dsh, float* dL_dcov3D,
                        void* stream);

/*
 * Batched entry points: K independent renders ("jobs") per call, ONE kernel launch per pipeline stage for up to
 * eight jobs (larger K is processed in groups of eight).  Two uses on ExAvatar's path:
 *   - the K training views one GPU holds of a view-sharded step (same Gaussian tensors, K cameras; the reference's
 *     per-sample loop avatar/main/model.py:81): exa_raster_backward_batch(..., sum_shared = 1) then returns the SUM
 *     of the K views' gradients in job 0's outputs (dL_dmeans2D stays per view);
 *   - the five same-camera renders of one iteration (scene, human, scene+human, human refined, scene+human refined;
 *     avatar/main/model.py:119-167): five jobs with their own Gaussian tensors, sum_shared = 0.
 * A single 1024x1024 render leaves most of the chip idle in every stage but the blend, so a batch costs far less
 * than K calls.  Results are bit-identical to K single calls (sum_shared: up to the order of the K-term sum).
 * All jobs of one call share `store_ctx`.  Same workspace / ownership / error rules as the single-render calls,
 * which are implemented as a batch of one.
 */
typedef struct ExaRasterForwardJob {
    const ExaRasterSettings* settings;
    int32_t P, sh_M;
    const float* means3D; const float* shs; const float* colors_precomp; const float* opacities;
    const float* scales; const float* rotations; const float* cov3D_precomp;
    int32_t* radii;                 /* [dev] out int32[P]                                              */
    void* geom_ws; void* tile_ws;   /* sized by exa_raster_workspace_sizes(P, W, H, capacity)          */
    void* bin_ws; uint64_t capacity;                     /* ignored by exa_raster_forward_bin_batch    */
    float* out_color; float* out_depth; float* out_alpha; /* ignored by exa_raster_forward_bin_batch   */
    int32_t keep_sorted_keys;       /* != 0: this render will be a SOURCE of a composite (exa_raster_forward_compose_batch):
                                       the sort keeps the sorted 64-bit keys (in place, no extra memory)      */
    /* Optional zero-copy header report (NULL = off): a DEVICE-VISIBLE address of 16 bytes of pinned host memory
     * (exa_raster_host_device_pointer).  As soon as the instance count of this job is known -- in the scatter stage,
     * long before the blend finishes -- ONE thread stores {num_rendered, overflow, num_visible, header_tag} there
     * (system-scope, the tag last).  A caller that polls the tag learns whether the capacity was enough without any
     * runtime call, copy or synchronisation; it works inside a captured hipGraph too (a plain store).  Use a fresh tag
     * (or reset the slot) per call to tell a new report from an old one. */
    void* host_header; uint32_t header_tag;
    /* Optional (NULL = off): [dev] out uint8[P], is_vis[i] = radii[i] > 0, written by the per-Gaussian kernel next to radii --
     * the `is_vis` of the reference's renderer (avatar/common/nets/layer.py: `radius > 0`) without a kernel of its own. */
    uint8_t* is_vis;
} ExaRasterForwardJob;

typedef struct ExaRasterBackwardJob {
    const ExaRasterSettings* settings;
    int32_t P, sh_M;
    const float* means3D; const float* shs; const float* colors_precomp; const float* opacities;
    const float* scales; const float* rotations; const float* cov3D_precomp;
    const int32_t* radii;
    const void* geom_ws; const void* tile_ws; const void* bin_ws; uint64_t capacity;
    const float* dL_dcolor; const float* dL_ddepth; const float* dL_dalpha;
    void* grad_ws;
    float* dL_dmeans2D; float* dL_dmeans3D; float* dL_dcolors; float* dL_dopacity;
    float* dL_dscales; float* dL_drotations; float* dL_dsh; float* dL_dcov3D;
    /* optional fused densification statistics (all three NULL = off): updated in place by the per-Gaussian backward
     * kernel exactly as exa_raster_densify_stats would do it with this job's dL_dmeans2D and radii -- no extra pass.
     * The update is a plain read-modify-write: two jobs of one batched call must not name the same array
     * (EXA_RASTER_E_ALIAS), with one exception -- sum_shared = 1 and the SAME three arrays in all K jobs: the K views'
     * statistics are then summed inside the kernel and written once per Gaussian. */
    float* densify_grad_accum; float* densify_track_cnt; float* densify_radius_max;
    /* Constant prefix: Gaussians 0 .. grad_first - 1 are inputs only, they get no gradient.  This is the scene under
     * the human in ExAvatar's composite renders (torch.cat((scene.detach(), human)), avatar/main/model.py:119-126): the
     * backward skips every 64-entry batch that blended no trainable Gaussian, writes no partial record for a constant one
     * and runs no per-Gaussian chain rule for it.  With grad_first > 0 EVERY gradient array above (dL_dmeans2D ...
     * dL_dcov3D, densify_*) holds P - grad_first rows: row r belongs to Gaussian grad_first + r.  0 = all trainable.
     * Not combinable with sum_shared. */
    int32_t grad_first;
    /* Composite render (exa_raster_forward_compose_batch): compose_geom_a != NULL makes this the backward of a composite.
     * Then P, the input tensors, radii and every gradient array describe source B (the trainable Gaussians), geom_ws is B's
     * splat workspace, tile_ws / bin_ws / capacity are the COMPOSITE's workspaces, grad_ws holds exa_raster_compose_sizes().grad_bytes (40 B x compose_capacity_b),
     * compose_geom_a / compose_P_a name source A's records (constants: no gradient) and grad_first must be 0. */
    const void* compose_geom_a; int32_t compose_P_a; uint64_t compose_capacity_b;
    /* Optional (NULL = off): DEVICE address of one pointer that the kernels load at EXECUTION time and read dL/dcolor from,
     * instead of `dL_dcolor` (which must still be non-NULL: it says that a colour gradient exists).  For callers that
     * replay this call from a captured hipGraph while the gradient arrives in a different tensor every time (autograd
     * hands `backward` a fresh one per iteration): exa_raster_store_pointers, enqueued on the same stream ahead of the
     * replay, points the call at it -- no 12-byte-per-pixel copy into a static buffer (GraphedIteration). */
    const float* const* dL_dcolor_indirect;
    /* != 0: dL_dmeans3D, dL_dcolors, dL_dsh, dL_dopacity, dL_dscales, dL_drotations and dL_dcov3D already HOLD gradients
     * (of another render of the same Gaussians, e.g. the composite that shows them over the scene) and this call ADDS its
     * own to them: out = held + this render's, one read more per value and no separate summation pass.  dL_dmeans2D is
     * per render and always overwritten.  Not combinable with sum_shared. */
    int32_t accumulate;
    /* Optional (0 = capacity / 64): how many 64-instance batch slots of the instance buffer the forward call actually used =
     * ceil(header.num_rendered / 64), which a caller that has seen the header (or its zero-copy report) knows by now.  The
     * backward blend launches one wave per batch slot; with a buffer sized generously (capacity = 1.5 x an earlier need, or
     * capacity_a + capacity_b for a composite, whose packed lists typically fill a fifth of that) most of those waves only
     * find out that their slot is past the end -- tens of thousands of dispatches (~0.24 ns each chip-wide, 24 us per
     * five-render iteration for the two composites).  A value below the true count would skip batches: pass only what the
     * header of THIS forward call said. */
    uint32_t used_slots;
} ExaRasterBackwardJob;

/* A schedule of per-frame parameter blocks that is RESIDENT on the device -- the ring of cameras of a turntable animation
 * (avatar/main/animate_view_rot.py:104 computes all of them up front), the views of a benchmark -- stepped through without any
 * host-side work per frame: one 64-lane launch copies row (*counter mod n_rows) of `table` [n_rows x row_floats] to `dst` and
 * advances the counter (device int32).  Capturable: as the first node of a hipGraph every replay renders the next row
 * (a camera block as exa_raster_camera_block writes it: viewmatrix 16 | projmatrix 16 | campos 3 floats), where an eager copy
 * kernel in front of the replay costs ~4.5 us of GPU time per frame (system-scope fences around a launch outside the graph). */
int exa_raster_select_row(const float* table, int32_t n_rows, int32_t row_floats, int32_t* counter, float* dst, void* stream);

/* Stores `n` (<= 16) pointers into `table` (device memory) with one tiny kernel, in stream order. */
int exa_raster_store_pointers(void* table, const void* const* ptrs, int32_t n, void* stream);

/*
 * Composite renders: "A and B rendered together" from two renders of the SAME camera and image size that already exist,
 * without preprocessing, binning or sorting anything again -- the sorted list of every sub-tile is the merge of the two
 * sources' sorted lists (order of the concatenation cat(A, B): ascending depth, ties by index, so A wins ties).
 * This is what ExAvatar's scene + human renders are (torch.cat((scene.detach(), human)), avatar/main/model.py:119-126,
 * next to plain renders of `scene` and `human` in the same iteration).  A is a constant of the backward
 * pass (the detached scene), B is trainable.  Results are bit-identical to rendering the concatenation.
 * Both sources must be finished forward calls on the same stream with keep_sorted_keys != 0 whose workspaces are still
 * alive; the composite owns tile_ws / bin_ws (exa_raster_compose_sizes) and its output images.  radii of the composite =
 * the sources' radii, A's first.
 */
typedef struct ExaRasterComposeJob {
    const ExaRasterSettings* settings;          /* image size as the sources'; bg of the composite                      */
    int32_t P_a, P_b;
    const void* geom_a; const void* tile_a; const void* bin_a; uint64_t capacity_a;
    const void* geom_b; const void* tile_b; const void* bin_b; uint64_t capacity_b;
    void* tile_ws; void* bin_ws; uint64_t capacity;      /* capacity_a + capacity_b always suffices                     */
    float* out_color; float* out_depth; float* out_alpha;
    void* host_header; uint32_t header_tag;             /* as in ExaRasterForwardJob                                    */
    /* Optional (all four NULL = off): source A's OWN finished output images and the background pointer it was rendered with.
     * Where B has no entry in a sub-tile, the composite's pixels ARE A's pixels -- same list, same arithmetic -- provided
     * the two backgrounds hold equal values (compared on the device): such sub-tiles then get an EMPTY list in the
     * composite (no merge, no batch slots, no backward waves) and their 64 pixels are copied from A's images instead of
     * being blended again.  In ExAvatar's scene + human composites (avatar/main/model.py:129,146: both on the default white
     * background) that is every sub-tile outside the person: ~3/4 of the image.  Bit-identical either way. */
    const float* a_color; const float* a_depth; const float* a_alpha; const float* a_bg;
    /* Optional (radii_out / is_vis_out NULL = off): the composite's radii [P_a + P_b] and is_vis [P_a + P_b] = the sources'
     * arrays one behind the other -- what a render of cat(A, B) returns (reference module.py:641-647) -- copied by the
     * workgroups of the ranges launch that zero-fill the workspace anyway: no launch of their own. */
    const int32_t* radii_a; const int32_t* radii_b; int32_t* radii_out;
    const uint8_t* is_vis_a; const uint8_t* is_vis_b; uint8_t* is_vis_out;
} ExaRasterComposeJob;
/* tile_bytes / bin_bytes of a composite's workspaces, grad_bytes of its backward scratch (geom_bytes = 0) */
int exa_raster_compose_sizes(int32_t W, int32_t H, uint64_t capacity, uint64_t capacity_b, ExaRasterWorkspaceSizes* out);
int exa_raster_forward_compose_batch(const ExaRasterComposeJob* jobs, int32_t K, int32_t store_ctx, void* stream);

/* `store_ctx` of exa_raster_forward_batch / exa_raster_forward_render_batch / exa_raster_forward_compose_batch: bit 0 = keep the
 * backward context (any caller that passes 0 / 1 is unaffected); the two stage bits split one call into two calls with the
 * SAME job array, so that the caller can put other work between the sorted lists and the blend -- e.g. run the list merges of
 * composite renders (which need their sources' sorted lists, not their images) on a second stream while the sources blend:
 *   ..._STAGE_NO_BLEND    everything up to the sorted per-sub-tile lists (composite: ranges + merged lists), no blend
 *   ..._STAGE_BLEND_ONLY  only the blend, on the lists an earlier NO_BLEND call with these jobs left in the workspaces
 * exa_raster_backward_batch takes the same two bits in its `sum_shared` argument (bit 0 = sum_shared): BLEND_ONLY = the blend's
 * backward (per-instance partial sums) without the per-Gaussian chain rule, NO_BLEND = only the chain rule, on the partial sums
 * an earlier BLEND_ONLY call with these jobs left in grad_ws -- e.g. to let the blend's backward of one batch overlap another
 * batch's on a second stream before a chain rule that adds to that batch's outputs (`accumulate`). */
#define EXA_RASTER_STORE_CTX        1
#define EXA_RASTER_STAGE_NO_BLEND   2
#define EXA_RASTER_STAGE_BLEND_ONLY 4
/* one more cut, in front of the sort (composite: in front of the list merges, which need the sources' sorted lists -- the
 * ranges before them only need the sources' binning): NO_SORT = everything before it, SORT_ONLY = only the sort / the merges */
#define EXA_RASTER_STAGE_NO_SORT    8
#define EXA_RASTER_STAGE_SORT_ONLY  16

int exa_raster_forward_bin_batch(const ExaRasterForwardJob* jobs, int32_t K, void* stream);
int exa_raster_forward_render_batch(const ExaRasterForwardJob* jobs, int32_t K, int32_t store_ctx, void* stream);
int exa_raster_forward_batch(const ExaRasterForwardJob* jobs, int32_t K, int32_t store_ctx, void* stream);
int exa_raster_backward_batch(const ExaRasterBackwardJob* jobs, int32_t K, int32_t sum_shared, void* stream);

/*
 * Asynchronous read-back of the first 16 bytes of the header (num_rendered, overflow, max_tile_list, num_visible) of a
 * tile workspace into caller-provided PINNED host memory (exa_raster_read_header_full_async: all sizeof(ExaRasterHeader)
 * = 28 bytes, dst needs 32), enqueued on `stream` behind the forward call: what a binding
 * does after a fused exa_raster_forward to learn -- later, without a host synchronisation now -- whether the capacity
 * was enough.  One runtime call instead of a slice + view + copy through the tensor library (~20 us of host time per
 * render in an eager training loop).  Not capturable in a hipGraph (a memcpy node): callers skip it under capture.
 */
int exa_raster_read_header_async(const void* tile_ws, void* host_dst16, void* stream);
int exa_raster_read_header_full_async(const void* tile_ws, void* host_dst32, void* stream);
/* Device-visible address of pinned (page-locked, host-coherent) memory at `host_ptr`, for ExaRasterForwardJob.host_header. */
int exa_raster_host_device_pointer(void* host_ptr, void** device_ptr_out);

/*
 * Camera block of GaussianRenderer.forward (module.py:604-608) from DEVICE-resident extrinsics: for R [dev float[9],
 * row-major 3x3] and t [dev float[3]] writes viewmatrix = [[R, t], [0, 0, 0, 1]]^T (row-major [4,4]), projmatrix =
 * viewmatrix @ proj^T and campos = -R^T t, the three device tensors ExaRasterSettings points at.  proj16_host = the
 * [4,4] of get_proj_matrix (transforms.py:43-64; a function of focal length and image size only) in HOST memory,
 * row-major, read during the call.  One tiny launch: a new camera per animation frame costs no read-back, no host
 * matrix code and no upload (hipGraph-capturable).
 * Optional focal-length check (both NULL = off): proj16_host and the tan(fov) of ExaRasterSettings were derived from a
 * focal length the caller remembers (fx_expected, fy_expected); `focal` = this frame's [dev float[2]].  The kernel
 * compares them and stores {equal ? 1 : 0, 0, 0, flag_tag} into the 16 bytes at `host_flag` (a device-visible address of
 * pinned host memory, exa_raster_host_device_pointer) -- a caller that is handed a fresh focal tensor with every frame
 * (a data loader) learns about a zoom without ever reading the tensor back.
 */
int exa_raster_camera_block(const float* R, const float* t, const float* proj16_host, float* viewmatrix_out,
                            float* projmatrix_out, float* campos_out, const float* focal, float fx_expected,
                            float fy_expected, void* host_flag, uint32_t flag_tag, void* stream);

/* upstream markVisible: present[i] = (view-space z of means3D[i] > 0.2). */
int exa_raster_mark_visible(const ExaRasterSettings* settings, int32_t P, const float* means3D,
                            uint8_t* present, void* stream);

/*
 * Fused densification statistics.  Replaces the PyTorch bookkeeping the reference runs after
 * every backward on the scene Gaussians -- avatar/main/train.py:49-54 (stack mean_2d.grad),
 * avatar/main/model.py:279-285 (radius_max), avatar/common/nets/module.py:155-157 (track_stats) -- whose
 * boolean-mask indexing costs a host synchronisation per statement.  For every Gaussian i with radii[i] > 0:
 *     xyz_grad_accum[i] += sqrt(g.x^2 + g.y^2)   with g = dL_dmeans2D[i] (the NDC-scaled screen gradient)
 *     track_cnt[i]      += 1
 *     radius_max[i]      = max(radius_max[i], (float)radii[i])
 * All arrays are device pointers of P elements ([P,3] for dL_dmeans2D); any of the three outputs may be NULL.
 * The same update can ride along in the backward pass itself (ExaRasterBackwardJob.densify_*): the screen-space gradient
 * is in registers there, so the statistics cost no extra P-sized pass.
 */
int exa_raster_densify_stats(int32_t P, const float* dL_dmeans2D, const int32_t* radii,
                             float* xyz_grad_accum, float* track_cnt, float* radius_max, void* stream);

/*
 * Fused SSIM map (the image-loss gradient producer in front of the rasterizer's backward).
 * Replaces class SSIM of reference avatar/common/nets/loss.py:31-74 -- five grouped 11x11 conv2d (Gaussian window,
 * sigma 1.5, zero padding 5) + elementwise ops + their autograd graph -- for N = batch * channels planes of H x W
 * floats.  forward writes ssim_map and, when the three dm_* pointers are non-NULL, the partial derivatives of
 * the map w.r.t. (mu1, E[x^2], E[x y]) that backward consumes; backward writes dL/d(img1) (the rendered image;
 * the target gets no gradient).  The mask / bbox options of the reference are plain tensor ops in the Python mirror
 * (exavatar_release_amd/losses.py).
 */
int exa_ssim_forward(int32_t N, int32_t H, int32_t W, const float* img1, const float* img2, float* ssim_map,
                     float* dm_dmu1, float* dm_dE11, float* dm_dE12, void* stream);
int exa_ssim_backward(int32_t N, int32_t H, int32_t W, const float* img1, const float* img2, const float* dL_dmap,
                      const float* dm_dmu1, const float* dm_dE11, const float* dm_dE12, float* dL_dimg1,
                      void* stream);

/*
 * Fused photometric loss of one render: the weighted L1 + (1 - SSIM) objective the reference builds from
 * class RGBLoss and class SSIM (avatar/common/nets/loss.py:11-74) at avatar/main/model.py:197-198, 204-205, 214-215
 *     loss = w_l1 * mean(l1_weight * |x - y|) + w_ssim * mean(1 - ssim(x * ssim_mask, y * ssim_mask))
 * over the crop window `crop` = {x0, y0, w, h} (the clamped bbox; the SSIM convolutions zero-pad at ITS border, as the
 * reference crops first, loss.py:50-58).  img_out / img_target: [B, C, H, W]; l1_weight, ssim_mask: [B, 1, H, W] or NULL.
 *   exa_photo_loss_forward  one kernel: SSIM statistics, the three partial-derivative maps (maps_ws: 3 * B * C * w * h
 *       floats) and per-workgroup partial sums (partials: 2 floats x exa_photo_loss_blocks(): sum of the SSIM map, sum
 *       of l1_weight |x - y|; the caller adds them up and forms the loss value).
 *   exa_photo_loss_grad     one kernel: dL/d(img_out) inside the crop window (the caller zero-fills dL_dimg outside it),
 *       ready to be handed to exa_raster_backward as dL_dcolor.
 * exa_l1_forward / exa_l1_backward: the L1 map of RGBLoss alone -- |x - t| over the crop with t = y * mask + (1 - mask) * bg
 * when mask [B,1,H,W] and bg [B,C] are given (loss.py:15-17) -- and sign(x - t) * dL_dmap.
 */
int64_t exa_photo_loss_blocks(int32_t B, int32_t C, int32_t crop_w, int32_t crop_h);
int exa_photo_loss_forward(int32_t B, int32_t C, int32_t H, int32_t W, const int32_t* crop, const float* img_out,
                           const float* img_target, const float* l1_weight, const float* ssim_mask, float* maps_ws,
                           float* partials, void* stream);
int exa_photo_loss_grad(int32_t B, int32_t C, int32_t H, int32_t W, const int32_t* crop, const float* img_out,
                        const float* img_target, const float* l1_weight, const float* ssim_mask, float w_l1, float w_ssim,
                        const float* maps_ws, float* dL_dimg, void* stream);
int exa_l1_forward(int32_t B, int32_t C, int32_t H, int32_t W, const int32_t* crop, const float* img_out,
                   const float* img_target, const float* mask, const float* bg, float* l1_map, void* stream);
int exa_l1_backward(int32_t B, int32_t C, int32_t H, int32_t W, const int32_t* crop, const float* img_out,
                    const float* img_target, const float* mask, const float* bg, const float* dL_dmap, float* dL_dimg,
                    void* stream);

/*
 * Optional per-kernel timing for benchmarks (the only state the library ever keeps, process-wide,
 * off by default, not thread-safe).  While enabled, every kernel / memset the library enqueues is bracketed by a
 * pair of hipEvents recorded on the caller's stream.  exa_raster_timing_read() synchronises on the
 * events of the most recent calls and returns their durations in milliseconds (-1 for a slot that
 * did not run), then clears the slots.  Slot names: exa_raster_timing_name(i), i < EXA_RASTER_TIMING_SLOTS.
 * Must not be enabled during hipGraph capture.
 */
#define EXA_RASTER_TIMING_SLOTS 9
int exa_raster_timing_enable(int32_t on);
int exa_raster_timing_read(float* ms_out, int32_t n);
const char* exa_raster_timing_name(int32_t slot);

#ifdef __cplusplus
}
#endif
#endif /* EXA_RASTER_H */
