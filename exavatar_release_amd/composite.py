"""Composite renders on the Python surface: "source A and source B rendered together" from two renders that already exist
(``exa_raster_forward_compose_batch``, csrc/compose.hip) -- ExAvatar's ``torch.cat((scene.detach(), human))`` renders
(``avatar/main/model.py:119-126``) without preprocessing, binning or sorting the concatenation.  Split out of ``rasterizer.py``
in round 6; the module-level state it shares with the plain renders (capture hooks, header pool, configuration) stays there and is
reached through ``rz``."""
import ctypes

import torch

from . import _lib
from . import rasterizer as rz


class _CJob:
    """Host-side record of one composite render."""
    __slots__ = ('a', 'b', 'rs', 'settings', 'keep', 'planes', 'radii', 'is_vis', 'ws', 'tile_ptr', 'bin_ptr', 'capacity', 'tb',
                 'report', 'key')


_side_streams = {}


def _compose_launch(cjobs, store_ctx, device, capturing, sorted_event=None):
    """Run the forward of the composite jobs against the workspaces of their sources.  ``sorted_event`` (inside a
    stream capture): recorded when the sources' sorted lists were complete, BEFORE their blend was queued -- ranges and list
    merges then run on a side stream concurrently with that blend, and the composites' own blend follows the join."""
    lib = _lib.load()
    K = len(cjobs)
    arr = (_lib.ExaRasterComposeJob * K)()
    stream_obj = torch.cuda.current_stream(device)
    pool = None if capturing else rz._pool()
    for k, c in enumerate(cjobs):
        ja, jb = c.a, c.b
        c.capacity = ja.capacity + jb.capacity
        sz = _lib.ExaRasterWorkspaceSizes()
        _lib.check(lib.exa_raster_compose_sizes(c.rs.image_width, c.rs.image_height, c.capacity, jb.capacity, ctypes.byref(sz)))
        c.tb = int(sz.tile_bytes)
        c.ws = rz._workspace(c.tb + int(sz.bin_bytes), device)
        c.tile_ptr = c.ws.data_ptr()
        c.bin_ptr = c.tile_ptr + c.tb
        a = arr[k]
        a.settings = ctypes.pointer(c.settings)
        a.P_a, a.P_b = ja.P, jb.P
        a.geom_a, a.tile_a, a.bin_a, a.capacity_a = ja.geom_ptr, ja.tile_ptr, ja.bin_ptr, ja.capacity
        a.geom_b, a.tile_b, a.bin_b, a.capacity_b = jb.geom_ptr, jb.tile_ptr, jb.bin_ptr, jb.capacity
        a.tile_ws, a.bin_ws, a.capacity = c.tile_ptr, c.bin_ptr, c.capacity
        base = c.planes.data_ptr()
        H, W = int(c.rs.image_height), int(c.rs.image_width)
        a.out_color, a.out_depth, a.out_alpha = base, base + 12 * H * W, base + 16 * H * W
        # source A's finished images: where B has no entry the composite's pixels are A's (equal backgrounds are checked on
        # the device): those sub-tiles skip merge, blend and backward (include/exa_raster.h, ExaRasterComposeJob.a_color)
        if rz.config.compose_reuse_source and ja.settings.bg:
            pa = ja.planes.data_ptr()
            a.a_color, a.a_depth, a.a_alpha, a.a_bg = pa, pa + 12 * H * W, pa + 16 * H * W, ja.settings.bg
        if c.radii is not None:
            a.radii_a, a.radii_b, a.radii_out = ja.radii.data_ptr(), jb.radii.data_ptr(), c.radii.data_ptr()
            a.is_vis_a, a.is_vis_b, a.is_vis_out = ja.is_vis.data_ptr(), jb.is_vis.data_ptr(), c.is_vis.data_ptr()
        c.report = None
        a.host_header, a.header_tag = None, 0
        if pool is not None:
            # (the composite's report is never waited for: its buffer holds both sources' capacities, so it cannot overflow
            #  once they did not; its backward reads the slot count from it if it has landed, rz._landed_need)
            slot, tag, dev_addr = pool.take()
            a.host_header, a.header_tag = dev_addr, tag
            c.report = (slot, tag)
        elif capturing and rz._capture_report_c is not None and k < len(rz._capture_report_c) and rz._capture_report_c[k] is not None:
            a.host_header = rz._hdr_pool.dev_base + 16 * rz._capture_report_c[k][0]
            a.header_tag = rz._capture_report_c[k][1]
    if sorted_event is None:
        _lib.check(lib.exa_raster_forward_compose_batch(arr, K, int(store_ctx), ctypes.c_void_p(stream_obj.cuda_stream)))
        return
    side = _side_streams.get(device.index)
    if side is None:
        side = _side_streams[device.index] = torch.cuda.Stream(device=device)
    ev_binned, ev_sorted = sorted_event
    if rz.config.poison:                   # (the 0xFF fill of the workspaces above was queued on THIS stream, behind the events)
        side.wait_stream(stream_obj)
    side.wait_event(ev_binned)          # ranges + zero-fill next to the sources' sort ...
    _lib.check(lib.exa_raster_forward_compose_batch(arr, K, int(store_ctx) | _lib.STAGE_NO_SORT, ctypes.c_void_p(side.cuda_stream)))
    side.wait_event(ev_sorted)          # ... list merges next to their blend
    _lib.check(lib.exa_raster_forward_compose_batch(arr, K, int(store_ctx) | _lib.STAGE_SORT_ONLY, ctypes.c_void_p(side.cuda_stream)))
    stream_obj.wait_stream(side)
    _lib.check(lib.exa_raster_forward_compose_batch(arr, K, int(store_ctx) | _lib.STAGE_BLEND_ONLY,
                                                    ctypes.c_void_p(stream_obj.cuda_stream)))


class _Compose(torch.autograd.Function):
    """K composite renders of pairs of finished renders (``exa_raster_forward_compose_batch``): render k shows source A
    (a constant: the detached scene) and source B (trainable: the human) together, from the sources' own splat records and
    sorted lists -- no preprocess, binning or sort of its own.  apply(K, sources, settings, grad_enabled, want_radii, token, *tensors[8 K]):
    ``sources[k] = (handle_a, handle_b)`` from ``rasterize_gaussians_batch(..., keep_keys=True)``; the tensors are B's inputs
    (the same objects its own render got), they receive this render's gradients.  ``token``: None, or the ``.token`` of the
    handles when ALL sources B come from that one batched call -- the gradients for B's tensors then travel through B's own
    backward (``rz.config.fold_composite_grads``) instead of being returned here."""

    @staticmethod
    def forward(ctx, K, sources, settings, grad_enabled, want_radii, token, *tensors):
        device = tensors[0].device
        need_ctx = bool(grad_enabled) and any(ctx.needs_input_grad[6:])      # (the token alone asks for nothing)
        capturing = torch.cuda.is_current_stream_capturing()
        cjobs = []
        with rz._on_device(device):
            for k in range(K):
                ja, jb = sources[k]
                rs = settings[k]
                if ja.device != device or jb.device != device:
                    raise ValueError('composite render: sources live on another device')
                if (ja.H, ja.W) != (jb.H, jb.W) or (int(rs.image_height), int(rs.image_width)) != (ja.H, ja.W):
                    raise ValueError('composite render: the sources and the composite must share one image size')
                for name in ('viewmatrix', 'projmatrix'):
                    if getattr(ja.settings, name) != getattr(jb.settings, name):
                        raise ValueError('composite render: the two sources were rendered with different cameras')
                if ja.nF or jb.nF:
                    raise ValueError('composite render: sources with a constant prefix are not supported')
                if not (ja.keep_keys and jb.keep_keys):
                    raise ValueError('composite render: sources must come from rasterize_gaussians_batch(..., keep_keys=True)')
                if tensors[rz.N_IN * k].shape[0] != jb.P:
                    raise ValueError('composite render: the tensors must be source B\'s inputs')
                c = _CJob()
                c.a, c.b, c.rs = ja, jb, rs
                c.keep = []
                c.settings = rz._make_settings(rs, device, c.keep)
                if c.settings.viewmatrix != ja.settings.viewmatrix:
                    raise ValueError('composite render: its camera differs from the sources\'')
                c.planes = torch.empty((5, ja.H, ja.W), dtype=rz._F32, device=device)
                # radii / is_vis of cat(A, B): written by the composite's own ranges launch (ExaRasterComposeJob.radii_out)
                c.radii = torch.empty(ja.P + jb.P, dtype=torch.int32, device=device) if want_radii else None
                c.is_vis = torch.empty(ja.P + jb.P, dtype=torch.bool, device=device) if want_radii else None
                c.key = ('compose', device.index, ja.P, jb.P, ja.H, ja.W)
                cjobs.append(c)
            _compose_launch(cjobs, need_ctx, device, capturing,
                            rz._overlap.pop(device.index, None) if capturing and rz.config.overlap_composites else None)
        ctx.need_ctx = need_ctx
        rz._tls.is_vis = [c.is_vis for c in cjobs]
        outs = []
        for c in cjobs:
            col, d, al = torch.split_with_sizes(c.planes, rz._PLANES)
            outs += [col, c.radii, d, al]
        if need_ctx:
            ctx.K, ctx.cjobs, ctx.device = K, cjobs, device
            ctx.fold = token is not None and bool(ctx.needs_input_grad[5]) and rz.config.fold_composite_grads and \
                all(c.b.token_ref is not None and c.b.token_ref() is token for c in cjobs)
            # B's converted inputs go through save_for_backward like _Rasterize's: an in-place update between the sources'
            # forward and this backward (the splat records of the sources hold the OLD values) raises instead of mixing
            saved, empty = [], None
            for c in cjobs:
                jb = c.b
                for t in (jb.means3D, jb.sh, jb.colors, jb.opac, jb.scales, jb.rot, jb.cov):
                    if t is None:
                        if empty is None:
                            empty = torch.empty(0, device=device)
                        t = empty
                    saved.append(t)
            ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(*[outs[4 * k + 1] for k in range(K) if outs[4 * k + 1] is not None])
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        if not ctx.need_ctx:
            raise RuntimeError('exavatar_release_amd: backward called on a composite that stored no context')
        lib = _lib.load()
        K, cjobs, device = ctx.K, ctx.cjobs, ctx.device
        saved = ctx.saved_tensors          # (checks the version counters of B's inputs)
        need = ctx.needs_input_grad[6:]
        arr = (_lib.ExaRasterBackwardJob * K)()
        keep, ret = [], [None, None, None, None, None, None]
        with rz._on_device(device):
            side, n_stashed = None, 0
            if ctx.fold and rz.config.overlap_composites and torch.cuda.is_current_stream_capturing():
                side = _side_streams.get(device.index)
                if side is None:
                    side = _side_streams[device.index] = torch.cuda.Stream(device=device)
            for k, c in enumerate(cjobs):
                ja, jb = c.a, c.b
                P, H, W, sh_M = jb.P, jb.H, jb.W, jb.sh_M
                has_sh, has_col, has_sc, has_rot, has_cov = [t is not None for t in (jb.sh, jb.colors, jb.scales, jb.rot, jb.cov)]
                g_color = rz._grad_in(grads[4 * k], (3, H, W), device)
                if g_color is None:
                    g_color = torch.zeros((3, H, W), dtype=rz._F32, device=device)
                g_depth = rz._grad_in(grads[4 * k + 2], (1, H, W), device)
                g_alpha = rz._grad_in(grads[4 * k + 3], (1, H, W), device)
                nd = need[rz.N_IN * k: rz.N_IN * (k + 1)]
                want = ((nd[0], 3), (nd[1], 3), (has_col and nd[3], 3), (nd[4], 1), (has_sc and nd[5], 3), (has_rot and nd[6], 4),
                        (has_cov and nd[7], 6))
                widths = [w for on, w in want if on]
                pieces = iter(torch.split_with_sizes(torch.empty(P * sum(widths), dtype=rz._F32, device=device), [P * w for w in widths])) \
                    if widths else iter(())
                d_means3D, d_means2D, d_colors, d_opac, d_scales, d_rot, d_cov = \
                    [next(pieces).view(P, w) if on else None for on, w in want]
                d_sh = torch.empty((P, sh_M, 3), dtype=rz._F32, device=device) if has_sh and nd[2] else None
                grad_ws = rz._workspace(rz._sizes(P, W, H, jb.capacity).grad_bytes, device)      # (B's Gaussian-major instance numbering)
                keep += [g_color, g_depth, g_alpha, grad_ws]
                a = arr[k]
                a.settings = ctypes.pointer(c.settings)
                a.P, a.sh_M = P, sh_M
                b_m3, b_sh, b_col, b_op, b_sc, b_rot, b_cov = saved[7 * k: 7 * k + 7]
                a.means3D = b_m3.data_ptr()
                a.shs = b_sh.data_ptr() if has_sh else None
                a.colors_precomp = b_col.data_ptr() if has_col else None
                a.opacities = b_op.data_ptr()
                a.scales = b_sc.data_ptr() if has_sc else None
                a.rotations = b_rot.data_ptr() if has_rot else None
                a.cov3D_precomp = b_cov.data_ptr() if has_cov else None
                a.radii = jb.radii.data_ptr()
                a.geom_ws, a.tile_ws, a.bin_ws, a.capacity = jb.geom_ptr, c.tile_ptr, c.bin_ptr, c.capacity
                a.dL_dcolor, a.dL_ddepth, a.dL_dalpha = g_color.data_ptr(), rz._addr(g_depth), rz._addr(g_alpha)
                if rz._capture_grad_ind is not None:
                    a.dL_dcolor_indirect = rz._capture_grad_ind.get(g_color.data_ptr())
                a.grad_ws = grad_ws.data_ptr()
                a.dL_dmeans2D, a.dL_dmeans3D, a.dL_dcolors = rz._addr(d_means2D), rz._addr(d_means3D), rz._addr(d_colors)
                a.dL_dopacity, a.dL_dscales, a.dL_drotations = rz._addr(d_opac), rz._addr(d_scales), rz._addr(d_rot)
                a.dL_dsh, a.dL_dcov3D = rz._addr(d_sh), rz._addr(d_cov)
                a.grad_first = 0
                a.compose_geom_a, a.compose_P_a, a.compose_capacity_b = ja.geom_ptr, ja.P, jb.capacity
                # the composite's packed lists fill a fraction of its buffer (sized for both sources): its own report, written
                # by the first kernel of its forward, says how many batch slots the backward has to visit
                need_c = rz._landed_need(c.report)
                a.used_slots = (need_c + 63) // 64 if need_c else 0
                if rz._capture_used is not None and k < len(rz._capture_used[1]):
                    a.used_slots = int(rz._capture_used[1][k])
                me = (id(ctx), k)
                if ctx.fold and (jb.stash is None or jb.stash[0] == me):
                    # leave them with B's job: B's own backward runs after this one (the token orders it) and adds its
                    # gradients to these buffers inside its per-Gaussian kernel, then returns them as the tensors' gradients
                    jb.stash = (me, rz._grad_pattern(nd[0], d_sh is not None, has_col and nd[3], nd[4], has_sc and nd[5],
                                                  has_rot and nd[6], has_cov and nd[7]),
                                (d_means3D, d_sh, d_colors, d_opac, d_scales, d_rot, d_cov)) + ((side,) if side is not None else ())
                    n_stashed += 1
                    ret += [None, d_means2D, None, None, None, None, None, None]
                    if ret[5] is None:
                        ret[5] = torch.empty(0, dtype=rz._F32, device=device)
                else:
                    ret += [d_means3D, d_means2D, d_sh, d_colors, d_opac, d_scales, d_rot, d_cov]
            if side is not None and n_stashed == K:
                # every gradient of this call travels through the sources' own backward, which joins the side stream before
                # it reads them: this whole backward overlaps the sources' blend backward (graph edges inside the capture)
                side.wait_stream(torch.cuda.current_stream(device))
                for t in keep:                    # scratch + incoming gradients: allocated on this stream, read on the other --
                    if torch.is_tensor(t):        # their memory must not be handed out again before the join
                        t.record_stream(side)
                _lib.check(lib.exa_raster_backward_batch(arr, K, 0, ctypes.c_void_p(side.cuda_stream)))
            else:
                _lib.check(lib.exa_raster_backward_batch(arr, K, 0, rz._stream_ptr(device)))
                if side is not None:              # (not all folded: nobody downstream would join the side stream)
                    for c in cjobs:
                        if c.b.stash is not None and len(c.b.stash) > 3:
                            c.b.stash = c.b.stash[:3]
        return tuple(ret)


def rasterize_composites(sources, jobs, token=None, radii=True):
    """K composite renders -- "source A and source B rendered together" -- from renders that already exist.

    ``sources``: K pairs ``(handle_a, handle_b)`` of handles returned by ``rasterize_gaussians_batch(..., keep_keys=True)``
    (same camera, same image size, same stream); A is treated as a constant (ExAvatar's detached scene,
    ``avatar/main/model.py:119-126``), B is trainable.  ``jobs``: K dicts with B's keyword tensors (the very tensors its own
    render got; they receive this render's gradients), a fresh ``means2D`` probe of B's length and ``raster_settings`` (the
    composite's background).  The composite reuses the sources' splat records and MERGES their sorted per-sub-tile lists:
    no preprocess, binning or sort of its own, bit-identical to rendering ``cat(A, B)``.  Returns K ``(color, radii, depth,
    alpha)`` tuples; ``radii`` = ``cat(radii_a, radii_b)`` (and ``take_is_vis()`` the matching ``is_vis``), copied by the
    composite's own first launch.  ``token``: the ``.token`` of the handles list when every source B
    belongs to that one batched call (see :class:`_Compose`); None is always correct.  ``radii=False``: the radii slot of
    the returned tuples is None (no concatenation kernel; the caller builds it from the sources' radii if anybody asks)."""
    jobs = list(jobs)
    K = len(jobs)
    if K == 0:
        return []
    if len(sources) != K:
        raise ValueError('rasterize_composites: one (handle_a, handle_b) pair per job')
    flat = []
    for j in jobs:
        flat += [j.get(n) for n in rz._IN_NAMES]
    outs = _Compose.apply(K, tuple(tuple(s) for s in sources), tuple(j['raster_settings'] for j in jobs),
                          torch.is_grad_enabled(), bool(radii), token, *flat)
    return [tuple(outs[4 * k: 4 * k + 4]) for k in range(K)]
