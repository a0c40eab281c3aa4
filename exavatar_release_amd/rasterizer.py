"""Drop-in Python surface of the rasterizer the reference imports at
``avatar/common/nets/module.py:11``::

    from diff_gaussian_rasterization_depth import GaussianRasterizationSettings, GaussianRasterizer

Same class names, the same 12-field settings tuple (module.py:609-622), the same keyword arguments
and the same ``(color, radii, depth, alpha)`` return order (module.py:632-640), implemented as a
``torch.autograd.Function`` over the C ABI of ``libexa_raster.so`` (hand-written HIP for gfx950).
Tensors stay PyTorch-ROCm tensors; only raw device pointers cross the boundary.

One autograd node serves K >= 1 renders ("jobs"): ``rasterize_gaussians`` is a batch of one,
``rasterize_gaussians_batch`` puts K renders -- K training views of the same Gaussians, or the five
same-camera renders of an ExAvatar iteration (``avatar/main/model.py:119-167``) -- into ONE launch per
pipeline stage (``exa_raster_*_batch``, include/exa_raster.h).

Instance-buffer sizing.  The number of (Gaussian, tile) instances D is only known on the device
after the binning stage.  Policies (``config.mode``):

* ``'exact'`` (what upstream does): stage 1, read D back (16-byte D2H copy, one stream sync),
  allocate exactly, stage 2.
* ``'capacity'``: one fused call with a buffer sized from the D of earlier calls of the same shape
  (x ``config.capacity_growth``; the first call of a shape runs in exact mode to measure D), or from
  ``config.fixed_capacity``; no host sync and hipGraph-capturable.  An overflow is latched on
  the device: the forward outputs of that call are invalid and its backward writes zero gradients.  It
  is surfaced (``RuntimeError``) by the render's own ``backward`` before any gradient is returned, by
  :func:`check_overflow`, and at the start of every later call once the asynchronous read-back has landed.
* ``'auto'`` (default): ``'capacity'`` for renders that will be differentiated (the training loop: the
  overflow check sits in ``backward``, so an optimizer step never sees gradients of an overflowed render)
  and under stream capture, ``'exact'`` for ``torch.no_grad()`` renders.
"""
import ctypes
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class _Config:
    mode = 'auto'             # 'auto' | 'exact' | 'capacity'
    capacity_growth = 1.5     # capacity mode: head-room over the largest D seen so far
    min_capacity = 1 << 16
    fixed_capacity = None     # capacity mode: use exactly this many instances (e.g. calibrated by a warm-up)
    keep_debug = False        # developer probes: keep the workspaces of the most recent forward reachable


config = _Config()

_debug_last = {}  # only filled when config.keep_debug (tools/): workspaces of the most recent forward
_seen_D = {}      # (device index, P, H, W) -> largest instance capacity a call of that shape needed
_pending = []     # capacity-mode calls without a backward: (event, pinned header rows, [(key, capacity)])


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _addr(t):
    return t.data_ptr() if t is not None else None


def _f32c(t, name, device):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a tensor' % name)
    if t.device != device:
        raise ValueError('%s is on %s, expected %s' % (name, t.device, device))
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_CTX = _NoCtx()


def _on_device(device):
    """``torch.cuda.device(device)`` only when it is not the current device already (the context manager costs ~10 us
    per entry, twice per render, in an eager training loop)."""
    return _NO_CTX if torch.cuda.current_device() == device.index else torch.cuda.device(device)


def _make_settings(rs, device, keep):
    """ctypes settings struct; tensors it points to are appended to ``keep`` so they stay alive."""
    s = _lib.ExaRasterSettings()
    s.image_height = int(rs.image_height)
    s.image_width = int(rs.image_width)
    s.tanfovx = float(rs.tanfovx)
    s.tanfovy = float(rs.tanfovy)
    s.scale_modifier = float(rs.scale_modifier)
    s.sh_degree = int(rs.sh_degree)
    s.prefiltered = int(bool(rs.prefiltered))
    s.debug = int(bool(rs.debug))
    for name in ('bg', 'viewmatrix', 'projmatrix', 'campos'):
        t = getattr(rs, name)
        if not (isinstance(t, torch.Tensor) and t.device == device and t.dtype == torch.float32 and t.is_contiguous()):
            if not isinstance(t, torch.Tensor):
                t = torch.as_tensor(t, dtype=torch.float32)
            t = t.to(device=device, dtype=torch.float32).contiguous()
        keep.append(t)
        setattr(s, name, t.data_ptr())
    return s


def _note_header(key, cap, D, overflow):
    _seen_D[key] = max(_seen_D.get(key, 0), D)
    if overflow:
        raise RuntimeError('exavatar_release_amd: tile-instance buffer overflow (needed %d, capacity %d); the outputs '
                           'of that render are invalid and its gradients are zero. Use config.mode="exact" or raise '
                           'config.capacity_growth.' % (D, cap))


def _drain_pending(block=False):
    """Process finished asynchronous header read-backs of capacity-mode calls."""
    global _pending
    rest, todo = [], []
    for item in _pending:
        ev = item[0]
        if block:
            ev.synchronize()
        (todo if ev.query() else rest).append(item)
    _pending = rest
    for ev, host, jobs in todo:
        vals = host[:len(jobs)].tolist()
        _hdr_release(ev, host)
        for (key, cap), row in zip(jobs, vals):
            _note_header(key, cap, row[0], row[1])


def check_overflow():
    """Wait for all outstanding capacity-mode calls and raise if any overflowed its buffer."""
    _drain_pending(block=True)


def check_overflow_quiet():
    """Drain the outstanding capacity-mode read-backs like :func:`check_overflow`, but only RECORD what they say
    (the instance counts feed the capacity memo) instead of raising: for callers that handle an overflow themselves."""
    try:
        _drain_pending(block=True)
    except RuntimeError:
        pass


def read_header(tile_ws):
    """(num_rendered, overflow, entries, num_visible, num_instances) of a tile workspace tensor (synchronises)."""
    return tuple(int(v) for v in tile_ws[:20].view(torch.int32).cpu())


def last_header():
    """Header of the most recent forward; needs ``config.keep_debug = True`` (developer probes only)."""
    return read_header(_debug_last['tile'])


_hdr_pool = []    # recycled (event, pinned [8, 4] int32 buffer) pairs of completed header read-backs


def _hdr_slot(K):
    if K <= 8 and _hdr_pool:
        return _hdr_pool.pop()
    return torch.cuda.Event(), torch.empty((max(K, 8), 4), dtype=torch.int32, pin_memory=True)


def _hdr_release(ev, host):
    if len(_hdr_pool) < 64 and host.shape[0] == 8:
        _hdr_pool.append((ev, host))


_size_cache = {}


def _sizes(P, W, H, capacity):
    """exa_raster_workspace_sizes, memoised (a ctypes round trip per call adds up in eager training loops)."""
    key = (P, W, H, capacity)
    sz = _size_cache.get(key)
    if sz is None:
        if len(_size_cache) > 256:
            _size_cache.clear()
        sz = _size_cache[key] = _lib.workspace_sizes(P, W, H, capacity)
    return sz


class _Job:
    """Host-side record of one render of a batch."""
    __slots__ = ('rs', 'P', 'nF', 'H', 'W', 'sh_M', 'key', 'means3D', 'sh', 'colors', 'opac', 'scales', 'rot', 'cov',
                 'settings', 'keep', 'planes', 'radii', 'ws', 'bins', 'geom_ptr', 'tile_ptr', 'bin_ptr', 'capacity',
                 'gb', 'tb')


N_IN = 8      # tensor arguments per job: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D


class _Rasterize(torch.autograd.Function):
    """K renders, one launch per pipeline stage.  apply(K, settings, grad_enabled, shared, densify, frozen, *tensors[8 K]).
    ``densify``: None or a K-list of None / (xyz_grad_accum, track_cnt, radius_max) tensors updated IN PLACE by that
    render's backward (fused densification statistics, include/exa_raster.h).
    ``frozen``: None or a K-list of None / 8-tuples in the order of ``_IN_NAMES``: a CONSTANT prefix of Gaussians that
    render k blends in front of / behind its own (the detached scene of ExAvatar's composite renders,
    ``avatar/main/model.py:119-126``).  The render sees ``cat(prefix, own)``; the backward returns gradients for the own
    rows only and skips all work on the prefix (``ExaRasterBackwardJob.grad_first``)."""

    @staticmethod
    def forward(ctx, K, settings, grad_enabled, shared, densify, frozen, *tensors):
        lib = _lib.load()
        device = tensors[0].device
        if device.type != 'cuda':
            raise RuntimeError('exavatar_release_amd: the rasterizer runs on a ROCm device only '
                               '(got %s); there is no CPU path' % device)
        # autograd is off inside forward(): the caller samples torch.is_grad_enabled() (a render under no_grad must
        # not pay for the backward context although GaussianRenderer's mean_2d probe requires grad)
        need_ctx = bool(grad_enabled) and any(ctx.needs_input_grad)
        jobs = []
        for k in range(K):
            m3, _m2, sh, col, op, sc, rot, cov = tensors[N_IN * k: N_IN * (k + 1)]
            j = _Job()
            j.rs = settings[k]
            j.H, j.W = int(j.rs.image_height), int(j.rs.image_width)
            fz = frozen[k] if frozen is not None else None
            j.nF = int(fz[0].shape[0]) if fz is not None else 0

            def inp(t, i, name):
                t = _f32c(t, name, device)
                if fz is None:
                    return t
                f = _f32c(fz[i], name + ' (constant prefix)', device)
                if (t is None) != (f is None):
                    raise ValueError('%s: the constant prefix must provide the same inputs as the render' % name)
                if t is None:
                    return None
                if f.shape[0] != j.nF or f.shape[1:] != t.shape[1:]:
                    raise ValueError('%s: constant prefix of shape %s does not fit %s' % (name, tuple(f.shape), tuple(t.shape)))
                return torch.cat((f, t))          # no autograd inside Function.forward: a plain copy kernel
            j.means3D = inp(m3, 0, 'means3D')
            j.P = int(j.means3D.shape[0])
            j.sh = inp(sh, 2, 'shs')
            j.colors = inp(col, 3, 'colors_precomp')
            j.opac = inp(op, 4, 'opacities')
            j.scales = inp(sc, 5, 'scales')
            j.rot = inp(rot, 6, 'rotations')
            j.cov = inp(cov, 7, 'cov3D_precomp')
            j.sh_M = int(j.sh.shape[1]) if j.sh is not None else 0
            j.key = (device.index, j.P, j.H, j.W)
            jobs.append(j)

        if len(_seen_D) > 4096:       # P changes with every densification step: keep the capacity memo bounded
            _seen_D.clear()             # (here, before any key of this call is looked up; cleared shapes re-measure once)
        mode = config.mode
        capturing = torch.cuda.is_current_stream_capturing()
        if mode == 'auto':
            mode = 'capacity' if (need_ctx or capturing) else 'exact'
        elif mode not in ('exact', 'capacity'):
            raise ValueError('config.mode must be "auto", "exact" or "capacity"')
        if mode == 'capacity' and config.fixed_capacity is None and any(j.key not in _seen_D for j in jobs):
            if capturing:
                raise RuntimeError('exavatar_release_amd: capacity mode needs config.fixed_capacity (or one '
                                   'earlier un-captured call of the same shape) before stream capture')
            mode = 'exact'            # first call of this shape: measure D once, like upstream does

        with _on_device(device):
            stream_obj = torch.cuda.current_stream(device)
            stream = ctypes.c_void_p(stream_obj.cuda_stream)
            if mode == 'capacity' and not capturing:
                _drain_pending()          # event queries are illegal during stream capture
            arr = (_lib.ExaRasterForwardJob * K)()
            for k, j in enumerate(jobs):
                j.keep = []
                j.settings = _make_settings(j.rs, device, j.keep)
                j.planes = torch.empty((5, j.H, j.W), dtype=torch.float32, device=device)   # colour | depth | alpha
                j.radii = torch.empty((j.P,), dtype=torch.int32, device=device)
                sz = _sizes(j.P, j.W, j.H, 0)
                j.gb, j.tb = int(sz.geom_bytes), int(sz.tile_bytes)
                j.bins = None
                if mode == 'capacity':
                    if config.fixed_capacity is not None:
                        cap = int(config.fixed_capacity)
                    else:
                        cap = max(int(_seen_D[j.key] * config.capacity_growth), config.min_capacity)
                    j.capacity = (cap + 63) // 64 * 64
                    # ONE arena per render: splat records | tile workspace | bin workspace
                    j.ws = torch.empty(j.gb + j.tb + int(_sizes(j.P, j.W, j.H, j.capacity).bin_bytes),
                                       dtype=torch.uint8, device=device)
                    j.bin_ptr = j.ws.data_ptr() + j.gb + j.tb
                else:
                    j.capacity = 0
                    j.ws = torch.empty(j.gb + j.tb, dtype=torch.uint8, device=device)
                    j.bin_ptr = None
                j.geom_ptr = j.ws.data_ptr()
                j.tile_ptr = j.geom_ptr + j.gb
                a = arr[k]
                a.settings = ctypes.pointer(j.settings)
                a.P, a.sh_M = j.P, j.sh_M
                a.means3D, a.shs, a.colors_precomp = _addr(j.means3D), _addr(j.sh), _addr(j.colors)
                a.opacities, a.scales, a.rotations = _addr(j.opac), _addr(j.scales), _addr(j.rot)
                a.cov3D_precomp = _addr(j.cov)
                a.radii = j.radii.data_ptr()
                a.geom_ws, a.tile_ws, a.bin_ws, a.capacity = j.geom_ptr, j.tile_ptr, j.bin_ptr, j.capacity
                base = j.planes.data_ptr()
                a.out_color, a.out_depth, a.out_alpha = base, base + 12 * j.H * j.W, base + 16 * j.H * j.W

            hdr_check = None
            if mode == 'exact':
                _lib.check(lib.exa_raster_forward_bin_batch(arr, K, stream))
                rows = [j.ws[j.gb:j.gb + 16].view(torch.int32) for j in jobs]
                hdr = (rows[0] if K == 1 else torch.stack(rows)).cpu().view(K, 4)        # D2H + sync, as upstream does
                for k, j in enumerate(jobs):
                    j.capacity = max(int(hdr[k, 0]), 64)          # header reports whole 64-instance batch slots
                    _seen_D[j.key] = max(_seen_D.get(j.key, 0), int(hdr[k, 0]))
                    j.bins = torch.empty(int(_sizes(j.P, j.W, j.H, j.capacity).bin_bytes), dtype=torch.uint8, device=device)
                    arr[k].bin_ws, arr[k].capacity = j.bins.data_ptr(), j.capacity
                _lib.check(lib.exa_raster_forward_render_batch(arr, K, int(need_ctx), stream))
            else:
                _lib.check(lib.exa_raster_forward_batch(arr, K, int(need_ctx), stream))
                if not capturing:
                    ev, host = _hdr_slot(K)                      # pinned buffer + event from a small pool
                    hp = host.data_ptr()
                    for k, j in enumerate(jobs):                 # one runtime call per header: no tensor-library ops
                        _lib.check(lib.exa_raster_read_header_async(j.tile_ptr, hp + 16 * k, stream))
                    ev.record(stream_obj)
                    hdr_check = (ev, host, [(j.key, j.capacity) for j in jobs])
                    if not need_ctx:
                        _pending.append(hdr_check)

        if config.keep_debug:
            j = jobs[-1]
            _debug_last['tile'] = j.ws[j.gb:j.gb + j.tb]
            _debug_last['geom'] = j.ws[:j.gb]
            _debug_last['capacity'] = j.capacity
        ctx.need_ctx = need_ctx
        outs = []
        for j in jobs:
            outs += [j.planes[0:3], j.radii, j.planes[3:4], j.planes[4:5]]
        if need_ctx:
            ctx.K = K
            ctx.densify = densify
            ctx.shared = bool(shared) and K > 1
            ctx.hdr_check = hdr_check
            ctx.meta = [(j.rs, j.P, j.H, j.W, j.sh_M, j.capacity, j.gb, j.tb, j.settings, j.keep, j.ws, j.bins,
                         tuple(t is not None for t in (j.sh, j.colors, j.scales, j.rot, j.cov)), j.nF) for j in jobs]
            saved = []
            empty = None
            for j in jobs:
                for t in (j.means3D, j.sh, j.colors, j.opac, j.scales, j.rot, j.cov):
                    if t is None:
                        if empty is None:
                            empty = torch.empty(0, device=device)
                        t = empty
                    saved.append(t)
                saved.append(j.radii)
            ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(*[outs[4 * k + 1] for k in range(K)])
        # outputs nobody differentiates (radii, and depth / alpha when the loss ignores them) reach backward as None
        # instead of freshly zero-filled 4 MB tensors: the kernels take a null pointer for "no gradient"
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        if not ctx.need_ctx:
            raise RuntimeError('exavatar_release_amd: backward called on a forward that stored no context')
        lib = _lib.load()
        K = ctx.K
        saved = ctx.saved_tensors
        device = saved[0].device
        f32 = dict(dtype=torch.float32, device=device)
        need = ctx.needs_input_grad[6:]
        arr = (_lib.ExaRasterBackwardJob * K)()
        keep, ret = [], [None, None, None, None, None, None]
        with _on_device(device):
            for k in range(K):
                rs, P, H, W, sh_M, cap, gb, tb, st, _skeep, ws, bins, has, nF = ctx.meta[k]
                Pg = P - nF                               # rows of every gradient array (constant prefix excluded)
                has_sh, has_col, has_sc, has_rot, has_cov = has
                means3D, sh, col, opac, scales, rot, cov, radii = saved[8 * k: 8 * k + 8]
                g_color, g_depth, g_alpha = grads[4 * k], grads[4 * k + 2], grads[4 * k + 3]

                def grad_in(g, shape):
                    if g is None:
                        return None
                    g = g.to(**f32).expand(shape)
                    return g if g.is_contiguous() else g.contiguous()
                g_color = grad_in(g_color, (3, H, W))
                if g_color is None:
                    g_color = torch.zeros((3, H, W), **f32)
                g_depth = grad_in(g_depth, (1, H, W))
                g_alpha = grad_in(g_alpha, (1, H, W))
                nd = need[N_IN * k: N_IN * (k + 1)]
                own = (not ctx.shared) or k == 0          # shared: job 0's outputs receive the sum over the K views
                # separate tensors on purpose: AccumulateGrad adopts a whole tensor as `.grad` without a copy, a view
                # of a shared buffer would be cloned
                d_means3D = torch.empty((Pg, 3), **f32) if own and nd[0] else None
                d_means2D = torch.empty((Pg, 3), **f32) if nd[1] else None
                d_sh = torch.empty((Pg, sh_M, 3), **f32) if own and has_sh and nd[2] else None
                d_colors = torch.empty((Pg, 3), **f32) if own and has_col and nd[3] else None
                d_opac = torch.empty((Pg, 1), **f32) if own and nd[4] else None
                d_scales = torch.empty((Pg, 3), **f32) if own and has_sc and nd[5] else None
                d_rot = torch.empty((Pg, 4), **f32) if own and has_rot and nd[6] else None
                d_cov = torch.empty((Pg, 6), **f32) if own and has_cov and nd[7] else None
                grad_ws = torch.empty(int(_sizes(P, W, H, cap).grad_bytes), dtype=torch.uint8, device=device)
                keep += [g_color, g_depth, g_alpha, grad_ws]
                a = arr[k]
                a.settings = ctypes.pointer(st)           # built in forward; its tensors are kept alive by ctx.meta
                a.P, a.sh_M = P, sh_M
                a.means3D = means3D.data_ptr()
                a.shs = sh.data_ptr() if has_sh else None
                a.colors_precomp = col.data_ptr() if has_col else None
                a.opacities = opac.data_ptr()
                a.scales = scales.data_ptr() if has_sc else None
                a.rotations = rot.data_ptr() if has_rot else None
                a.cov3D_precomp = cov.data_ptr() if has_cov else None
                a.radii = radii.data_ptr()
                a.geom_ws = ws.data_ptr()
                a.tile_ws = ws.data_ptr() + gb
                a.bin_ws = bins.data_ptr() if bins is not None else ws.data_ptr() + gb + tb
                a.capacity = cap
                a.dL_dcolor, a.dL_ddepth, a.dL_dalpha = g_color.data_ptr(), _addr(g_depth), _addr(g_alpha)
                a.grad_ws = grad_ws.data_ptr()
                a.dL_dmeans2D, a.dL_dmeans3D, a.dL_dcolors = _addr(d_means2D), _addr(d_means3D), _addr(d_colors)
                a.dL_dopacity, a.dL_dscales, a.dL_drotations = _addr(d_opac), _addr(d_scales), _addr(d_rot)
                a.dL_dsh, a.dL_dcov3D = _addr(d_sh), _addr(d_cov)
                dens = ctx.densify[k] if ctx.densify is not None else None
                if dens is not None:
                    if d_means2D is None:          # the statistics need the screen-space gradient: compute it anyway
                        d_tmp = torch.empty((Pg, 3), **f32)
                        keep.append(d_tmp)
                        a.dL_dmeans2D = d_tmp.data_ptr()
                    a.densify_grad_accum, a.densify_track_cnt, a.densify_radius_max = [_addr(t) for t in dens]
                a.grad_first = nF
                ret += [d_means3D, d_means2D, d_sh, d_colors, d_opac, d_scales, d_rot, d_cov]
            _lib.check(lib.exa_raster_backward_batch(arr, K, int(ctx.shared), _stream_ptr(device)))
        if ctx.hdr_check is not None:
            # capacity mode: make sure this render's forward did not overflow BEFORE handing gradients to the optimizer
            # (an overflowed forward gets zero gradients from the kernels above).  The backward kernels are already
            # queued, so waiting for the forward's 16-byte header read-back does not idle the GPU.
            ev, host, jobs = ctx.hdr_check
            ctx.hdr_check = None
            ev.synchronize()
            vals = host[:len(jobs)].tolist()
            _hdr_release(ev, host)
            for (key, cap), row in zip(jobs, vals):
                _note_header(key, cap, row[0], row[1])
        return tuple(ret)


def _check_densify(dens, P, device):
    if dens is None:
        return None
    dens = tuple(dens)
    if len(dens) != 3:
        raise ValueError('densify_stats = (xyz_grad_accum, track_cnt, radius_max), any of them None')
    for name, t in zip(('xyz_grad_accum', 'track_cnt', 'radius_max'), dens):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != P or t.device != device):
            raise ValueError('%s must be a contiguous float32 tensor with %d elements on %s' % (name, P, device))
    return dens


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, densify_stats=None):
    """``densify_stats``: optional ``(xyz_grad_accum, track_cnt, radius_max)`` float32 tensors of P elements that THIS
    render's backward updates in place (fused densification statistics, see ``densify.track_densify_stats``)."""
    dens = None if densify_stats is None else [_check_densify(densify_stats, int(means3D.shape[0]), means3D.device)]
    return _Rasterize.apply(1, (raster_settings,), torch.is_grad_enabled(), False, dens, None, means3D, means2D, sh,
                            colors_precomp, opacities, scales, rotations, cov3Ds_precomp)


_IN_NAMES = ('means3D', 'means2D', 'shs', 'colors_precomp', 'opacities', 'scales', 'rotations', 'cov3D_precomp')


def rasterize_gaussians_batch(jobs):
    """K renders in one launch per pipeline stage.

    ``jobs``: sequence of dicts with the keyword arguments of ``GaussianRasterizer.forward`` plus
    ``raster_settings`` (and optionally ``densify_stats``, see :func:`rasterize_gaussians`, and ``frozen``: a dict with
    the same tensor keywords holding a CONSTANT prefix of Gaussians -- the render blends ``cat(prefix, own)``, ``radii``
    covers both, gradients / ``means2D`` / ``densify_stats`` cover the job's own Gaussians only and the backward does
    no work for the prefix).  Returns a list of ``(color, radii, depth, alpha)`` tuples, bit-identical to K single
    calls.  When every job passes the SAME tensor objects for the Gaussians (K views of one model), the backward
    sums the K views' gradients inside the per-Gaussian kernel (one thread walks the K views) instead of letting
    autograd add K gradient tensors.
    """
    jobs = list(jobs)
    K = len(jobs)
    if K == 0:
        return []
    flat = []
    frozen = None
    for j in jobs:
        _check_combo(j.get('shs'), j.get('colors_precomp'), j.get('scales'), j.get('rotations'), j.get('cov3D_precomp'))
        flat += [j.get(n) for n in _IN_NAMES]
    if any(j.get('frozen') is not None for j in jobs):
        frozen = []
        for j in jobs:
            fz = j.get('frozen')
            if fz is not None:
                if fz.get('means3D') is None:
                    raise ValueError('frozen: the constant prefix needs means3D')
                fz = tuple(None if n == 'means2D' or fz.get(n) is None else fz[n].detach() for n in _IN_NAMES)
            frozen.append(fz)
    shared = K > 1 and frozen is None and \
        all(all(jobs[k].get(n) is jobs[0].get(n) for n in _IN_NAMES if n != 'means2D') for k in range(1, K))
    if shared and K > 8:
        shared = False
    dens = None
    if any(j.get('densify_stats') is not None for j in jobs):
        dens = [_check_densify(j.get('densify_stats'), int(j['means3D'].shape[0]), j['means3D'].device) for j in jobs]
    outs = _Rasterize.apply(K, tuple(j['raster_settings'] for j in jobs), torch.is_grad_enabled(), shared, dens, frozen,
                            *flat)
    return [tuple(outs[4 * k: 4 * k + 4]) for k in range(K)]


def _check_combo(shs, colors_precomp, scales, rotations, cov3D_precomp):
    if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')


class GaussianRasterizer(nn.Module):
    """Same constructor / ``forward`` / ``markVisible`` surface as the third-party class the
    reference instantiates at module.py:623."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        lib = _lib.load()
        rs = self.raster_settings
        device = positions.device
        if device.type != 'cuda':
            raise RuntimeError('exavatar_release_amd: ROCm device tensors only')
        with torch.no_grad(), torch.cuda.device(device):
            pos = _f32c(positions, 'positions', device)
            keep = []
            st = _make_settings(rs, device, keep)
            out = torch.empty(pos.shape[0], dtype=torch.uint8, device=device)
            _lib.check(lib.exa_raster_mark_visible(ctypes.byref(st), int(pos.shape[0]), _ptr(pos), _ptr(out),
                                                   _stream_ptr(device)))
        return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        _check_combo(shs, colors_precomp, scales, rotations, cov3D_precomp)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings)
