"""Drop-in Python surface of the rasterizer the reference imports at
``avatar/common/nets/module.py:11``::

    from diff_gaussian_rasterization_depth import GaussianRasterizationSettings, GaussianRasterizer

Same class names, the same 12-field settings tuple (module.py:609-622), the same keyword arguments
and the same ``(color, radii, depth, alpha)`` return order (module.py:632-640), implemented as a
``torch.autograd.Function`` over the C ABI of ``libexa_raster.so`` (hand-written HIP for gfx950).
Tensors stay PyTorch-ROCm tensors; only raw device pointers cross the boundary.

Instance-buffer sizing.  The number of (Gaussian, tile) instances D is only known on the device
after the binning stage.  Two policies (``config.mode``):

* ``'exact'`` (default, what upstream does): stage 1, read D back (16-byte D2H copy, one stream
  sync), allocate exactly, stage 2.
* ``'capacity'``: one fused call with a buffer sized from the D of earlier calls of the same shape
  (x ``config.capacity_growth``; the first call of a shape runs in exact mode to measure D), or from
  ``config.fixed_capacity``; no host sync and hipGraph-capturable.  An overflow is latched on
  the device, surfaced by :func:`check_overflow` (also called at the start of every later call
  once the asynchronous read-back has landed) and raises ``RuntimeError``.
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
    mode = 'exact'            # 'exact' | 'capacity'
    capacity_growth = 1.5     # capacity mode: head-room over the largest D seen so far
    min_capacity = 1 << 16
    fixed_capacity = None     # capacity mode: use exactly this many instances (e.g. calibrated by a warm-up)


config = _Config()

# capacity-mode state, per (device index, P, H, W): largest D observed, pending async read-backs
_debug_last = {}   # tile workspace / capacity of the most recent forward (developer introspection only)
_seen_D = {}
_pending = []     # list of (event, pinned header tensor, key, capacity)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _f32c(t, name, device):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a tensor' % name)
    if t.device != device:
        raise ValueError('%s is on %s, expected %s' % (name, t.device, device))
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


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
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t, dtype=torch.float32)
        t = t.to(device=device, dtype=torch.float32).contiguous()
        keep.append(t)
        setattr(s, name, t.data_ptr())
    return s


def _drain_pending(block=False):
    """Process finished asynchronous header read-backs of capacity-mode calls."""
    global _pending
    rest = []
    for ev, host, key, cap in _pending:
        if block:
            ev.synchronize()
        if ev.query():
            D, overflow = int(host[0]), int(host[1])
            _seen_D[key] = max(_seen_D.get(key, 0), D)
            if overflow:
                _pending = [p for p in _pending if p[0] is not ev]
                raise RuntimeError('exavatar_release_amd: tile-instance buffer overflow (needed %d, capacity %d); '
                                   'the outputs of that call are invalid. Use config.mode="exact" or raise '
                                   'config.capacity_growth.' % (D, cap))
        else:
            rest.append((ev, host, key, cap))
    _pending = rest


def check_overflow():
    """Wait for all outstanding capacity-mode calls and raise if any overflowed its buffer."""
    _drain_pending(block=True)


def last_header():
    """(num_rendered, overflow, entries, num_visible, num_instances) of the most recent forward (synchronises)."""
    return tuple(int(v) for v in _debug_last['tile'][:20].view(torch.int32).cpu())


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


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        lib = _lib.load()
        device = means3D.device
        if device.type != 'cuda':
            raise RuntimeError('exavatar_release_amd: the rasterizer runs on a ROCm device only '
                               '(got %s); there is no CPU path' % device)
        rs = raster_settings
        H, W = int(rs.image_height), int(rs.image_width)
        P = int(means3D.shape[0])
        means3D = _f32c(means3D, 'means3D', device)
        sh = _f32c(sh, 'shs', device)
        colors_precomp = _f32c(colors_precomp, 'colors_precomp', device)
        opacities = _f32c(opacities, 'opacities', device)
        scales = _f32c(scales, 'scales', device)
        rotations = _f32c(rotations, 'rotations', device)
        cov3Ds_precomp = _f32c(cov3Ds_precomp, 'cov3D_precomp', device)
        sh_M = int(sh.shape[1]) if sh is not None else 0
        need_ctx = any(ctx.needs_input_grad)

        keep = []
        with torch.cuda.device(device):
            st = _make_settings(rs, device, keep)
            stream = _stream_ptr(device)
            u8 = dict(dtype=torch.uint8, device=device)
            planes = torch.empty((5, H, W), dtype=torch.float32, device=device)      # one allocation, three views
            color, depth, alpha = planes[0:3], planes[3:4], planes[4:5]
            radii = torch.empty((P,), dtype=torch.int32, device=device)
            sz = _sizes(P, W, H, 0)
            geom = torch.empty(int(sz.geom_bytes), **u8)
            tile = torch.empty(int(sz.tile_bytes), **u8)
            img = torch.empty(int(sz.img_bytes) if need_ctx else 0, **u8)
            inputs = (_ptr(means3D), _ptr(sh), _ptr(colors_precomp), _ptr(opacities), _ptr(scales), _ptr(rotations),
                      _ptr(cov3Ds_precomp))
            mode = config.mode
            if mode not in ('exact', 'capacity'):
                raise ValueError('config.mode must be "exact" or "capacity"')
            key = (device.index, P, H, W)
            capturing = torch.cuda.is_current_stream_capturing()
            if mode == 'capacity' and config.fixed_capacity is None and key not in _seen_D:
                if capturing:
                    raise RuntimeError('exavatar_release_amd: capacity mode needs config.fixed_capacity (or one '
                                       'earlier un-captured call of the same shape) before stream capture')
                mode = 'exact'            # first call of this shape: measure D once, like upstream does
            if mode == 'exact':
                _lib.check(lib.exa_raster_forward_bin(ctypes.byref(st), P, sh_M, *inputs, _ptr(radii), _ptr(geom),
                                                      _ptr(tile), stream))
                hdr = tile[:16].view(torch.int32).cpu()          # D2H + sync, as upstream does
                capacity = max(int(hdr[0]), 64)          # header reports whole 64-instance batch slots
                _seen_D[key] = max(_seen_D.get(key, 0), int(hdr[0]))
                bins = torch.empty(int(_sizes(P, W, H, capacity).bin_bytes), **u8)
                _lib.check(lib.exa_raster_forward_render(ctypes.byref(st), P, _ptr(geom), _ptr(tile), _ptr(bins),
                                                         capacity, _ptr(img), _ptr(color), _ptr(depth), _ptr(alpha),
                                                         int(need_ctx), stream))
            else:
                if not capturing:
                    _drain_pending()          # event queries are illegal during stream capture
                if config.fixed_capacity is not None:
                    capacity = int(config.fixed_capacity)
                else:
                    capacity = max(int(_seen_D[key] * config.capacity_growth), config.min_capacity)
                capacity = (capacity + 63) // 64 * 64
                bins = torch.empty(int(_sizes(P, W, H, capacity).bin_bytes), **u8)
                _lib.check(lib.exa_raster_forward(ctypes.byref(st), P, sh_M, *inputs, _ptr(radii), _ptr(geom),
                                                  _ptr(tile), _ptr(bins), capacity, _ptr(img), _ptr(color),
                                                  _ptr(depth), _ptr(alpha), int(need_ctx), stream))
                if not capturing:
                    host = torch.empty(4, dtype=torch.int32, pin_memory=True)
                    host.copy_(tile[:16].view(torch.int32), non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(device))
                    _pending.append((ev, host, key, capacity))

        _debug_last['tile'] = tile
        _debug_last['geom'] = geom
        _debug_last['capacity'] = capacity
        ctx.raster_settings = rs
        ctx.need_ctx = need_ctx
        if need_ctx:
            ctx.sh_M = sh_M
            ctx.capacity = capacity
            ctx.keep = keep
            ctx.settings_struct = st
            ctx.has = tuple(t is not None for t in (sh, colors_precomp, scales, rotations, cov3Ds_precomp))
            empty = torch.empty(0, device=device)
            ctx.save_for_backward(means3D, sh if sh is not None else empty,
                                  colors_precomp if colors_precomp is not None else empty, opacities,
                                  scales if scales is not None else empty,
                                  rotations if rotations is not None else empty,
                                  cov3Ds_precomp if cov3Ds_precomp is not None else empty,
                                  radii, geom, tile, bins, img)
        ctx.mark_non_differentiable(radii)
        # outputs nobody differentiates (radii, and depth / alpha when the loss ignores them) reach backward as None
        # instead of freshly zero-filled 4 MB tensors: the kernels take a null pointer for "no gradient"
        ctx.set_materialize_grads(False)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        if not ctx.need_ctx:
            raise RuntimeError('exavatar_release_amd: backward called on a forward that stored no context')
        lib = _lib.load()
        rs = ctx.raster_settings
        (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii, geom, tile, bins,
         img) = ctx.saved_tensors
        has_sh, has_col, has_sc, has_rot, has_cov = ctx.has
        device = means3D.device
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        f32 = dict(dtype=torch.float32, device=device)

        def grad_in(g, shape):
            if g is None:
                return None
            g = g.to(**f32).expand(shape).contiguous()
            return g
        g_color = grad_in(grad_color, (3, H, W))
        if g_color is None:
            g_color = torch.zeros((3, H, W), **f32)
        g_depth = grad_in(grad_depth, (1, H, W))
        g_alpha = grad_in(grad_alpha, (1, H, W))

        with torch.cuda.device(device):
            st = ctx.settings_struct          # built in forward; the tensors it points to are kept alive by ctx.keep
            # separate tensors on purpose: AccumulateGrad adopts a whole tensor as `.grad` without a copy, a view of a
            # shared buffer would be cloned
            d_means3D = torch.empty((P, 3), **f32)
            d_means2D = torch.empty((P, 3), **f32)
            d_opac = torch.empty((P, 1), **f32)
            d_colors = torch.empty((P, 3), **f32)
            d_scales = torch.empty((P, 3), **f32) if has_sc else None
            d_rot = torch.empty((P, 4), **f32) if has_rot else None
            d_sh = torch.empty((P, ctx.sh_M, 3), **f32) if has_sh else None
            d_cov = torch.empty((P, 6), **f32) if has_cov else None
            sz = _sizes(P, W, H, ctx.capacity)
            grad_ws = torch.empty(int(sz.grad_bytes), dtype=torch.uint8, device=device)
            _lib.check(lib.exa_raster_backward(
                ctypes.byref(st), P, ctx.sh_M,
                _ptr(means3D), _ptr(sh if has_sh else None), _ptr(colors_precomp if has_col else None),
                _ptr(opacities), _ptr(scales if has_sc else None), _ptr(rotations if has_rot else None),
                _ptr(cov3Ds_precomp if has_cov else None), _ptr(radii), _ptr(geom), _ptr(tile), _ptr(bins),
                ctx.capacity, _ptr(img), _ptr(g_color), _ptr(g_depth), _ptr(g_alpha), _ptr(grad_ws),
                _ptr(d_means2D), _ptr(d_means3D), _ptr(d_colors), _ptr(d_opac), _ptr(d_scales), _ptr(d_rot),
                _ptr(d_sh), _ptr(d_cov), _stream_ptr(device)))
        return (d_means3D, d_means2D, d_sh, d_colors if has_col else None, d_opac, d_scales, d_rot, d_cov, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


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
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings)
