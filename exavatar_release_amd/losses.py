"""Image losses in front of the rasterizer's backward: the producers of ``dL/d(render)``.

* ``PhotometricLoss`` -- the weighted L1 + (1 - SSIM) objective the reference assembles per render at
  ``avatar/main/model.py:197-198, 204-205, 214-215`` from ``RGBLoss`` and ``SSIM``
  (``avatar/common/nets/loss.py:11-74``), as TWO HIP kernels (``csrc/ssim.hip``: statistics + partial sums, then the
  gradient) that write ``dL/d(image)`` once, in the layout ``exa_raster_backward`` reads -- instead of five grouped
  11x11 ``conv2d``, ~25 elementwise / reduction kernels and their autograd graph.
* ``SSIM``     -- drop-in for the reference class (same constructor / ``forward`` signature and result): the map as one
  fused kernel, its backward as one more; differentiable in both images (SSIM is symmetric, the target's gradient is the
  same kernel with the roles swapped).
* ``RGBLoss``  -- drop-in for the reference class: the L1 map with optional mask / background / bbox as one kernel
  (and one for its backward) instead of a chain of elementwise ops.

ROCm device tensors only (no CPU path); the CPU restatements that pin these kernels live in ``oracle/loss_oracle.py``.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from .rasterizer import _ptr, _stream_ptr


def _crop_window(bbox, img_height, img_width):
    """(x0, y0, w, h) of the reference's bbox clamp (loss.py:19-24 / 51-56): top-left clamped at 0, bottom-right at the
    image size; the whole image when ``bbox`` is None."""
    if bbox is None:
        return 0, 0, img_width, img_height
    xmin, ymin, width, height = [int(v) for v in bbox[0]]
    x0, y0 = max(xmin, 0), max(ymin, 0)
    x1, y1 = min(x0 + width, img_width), min(y0 + height, img_height)
    return x0, y0, max(x1 - x0, 0), max(y1 - y0, 0)


def _c_crop(crop):
    return (ctypes.c_int32 * 4)(*crop)


def _dev_f32(t, device, name):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a tensor' % name)
    t = t.detach().to(device=device, dtype=torch.float32)
    return t if t.is_contiguous() else t.contiguous()


def _need_rocm(device, what):
    if device.type != 'cuda':
        raise RuntimeError('exavatar_release_amd: %s runs on a ROCm device only (no CPU path)' % what)


class _L1Map(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img_out, img_target, mask, bg, crop):
        lib = _lib.load()
        device = img_out.device
        _need_rocm(device, 'RGBLoss')
        if img_out.dim() != 4 or img_out.shape != img_target.shape:
            raise ValueError('RGBLoss expects two [B, C, H, W] images of the same shape')
        x, y = _dev_f32(img_out, device, 'img_out'), _dev_f32(img_target, device, 'img_target')
        B, C, H, W = x.shape
        compose = mask is not None and bg is not None
        m = _dev_f32(mask, device, 'mask').expand(B, 1, H, W).contiguous() if compose else None
        b = _dev_f32(bg, device, 'bg').expand(B, C).contiguous() if compose else None
        out = torch.empty((B, C, crop[3], crop[2]), dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            _lib.check(lib.exa_l1_forward(B, C, H, W, _c_crop(crop), _ptr(x), _ptr(y), _ptr(m), _ptr(b), _ptr(out),
                                          _stream_ptr(device)))
        ctx.crop = crop
        ctx.full = crop == (0, 0, W, H)
        ctx.save_for_backward(x, y, m if compose else x.new_empty(0), b if compose else x.new_empty(0))
        ctx.compose = compose
        return out

    @staticmethod
    def backward(ctx, grad_map):
        lib = _lib.load()
        x, y, m, b = ctx.saved_tensors
        B, C, H, W = x.shape
        g = grad_map.to(torch.float32).expand(B, C, ctx.crop[3], ctx.crop[2]).contiguous()
        dx = torch.empty_like(x) if ctx.full else torch.zeros_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.exa_l1_backward(B, C, H, W, _c_crop(ctx.crop), _ptr(x), _ptr(y), _ptr(m if ctx.compose else None),
                                           _ptr(b if ctx.compose else None), _ptr(g), _ptr(dx), _stream_ptr(x.device)))
        # d|x - t|/dt = -d|x - t|/dx; with a composed target t = y * mask + ..., dt/dy = mask
        dy = None
        if ctx.needs_input_grad[1]:
            dy = -dx * m if ctx.compose else -dx
        return dx if ctx.needs_input_grad[0] else None, dy, None, None, None


class RGBLoss(nn.Module):
    """``forward(img_out, img_target, bbox=None, mask=None, bg=None)`` -> L1 map, as reference loss.py:11-29."""

    def __init__(self):
        super(RGBLoss, self).__init__()

    def forward(self, img_out, img_target, bbox=None, mask=None, bg=None):
        crop = _crop_window(bbox, img_out.shape[2], img_out.shape[3])
        return _L1Map.apply(img_out, img_target, mask, bg, crop)


class _FusedSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img_out, img_target):
        lib = _lib.load()
        device = img_out.device
        _need_rocm(device, 'the fused SSIM')
        x = img_out.detach().to(torch.float32).contiguous()
        y = img_target.detach().to(device=device, dtype=torch.float32).contiguous()
        if x.dim() != 4 or x.shape != y.shape:
            raise ValueError('SSIM expects two [B, C, H, W] images of the same shape')
        B, C, H, W = x.shape
        # (autograd is off in here: needs_input_grad is what tells whether a backward can follow)
        need_x, need_y = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        out = torch.empty_like(x)
        maps_x = [torch.empty_like(x) for _ in range(3)] if need_x else [None, None, None]
        with torch.cuda.device(device):
            _lib.check(lib.exa_ssim_forward(B * C, H, W, _ptr(x), _ptr(y), _ptr(out), _ptr(maps_x[0]), _ptr(maps_x[1]),
                                            _ptr(maps_x[2]), _stream_ptr(device)))
            maps_y = [None, None, None]
            if need_y:        # SSIM is symmetric: the target's derivative maps are those of ssim(y, x)
                maps_y = [torch.empty_like(x) for _ in range(3)]
                scratch = torch.empty_like(x)
                _lib.check(lib.exa_ssim_forward(B * C, H, W, _ptr(y), _ptr(x), _ptr(scratch), _ptr(maps_y[0]),
                                                _ptr(maps_y[1]), _ptr(maps_y[2]), _stream_ptr(device)))
        ctx.need = (need_x, need_y)
        e = x.new_empty(0)
        ctx.save_for_backward(x, y, *[m if m is not None else e for m in maps_x + maps_y])
        return out

    @staticmethod
    def backward(ctx, grad_map):
        lib = _lib.load()
        x, y, a0, a1, a2, b0, b1, b2 = ctx.saved_tensors
        B, C, H, W = x.shape
        g = grad_map.to(torch.float32).expand(x.shape).contiguous()
        dx = dy = None
        with torch.cuda.device(x.device):
            if ctx.need[0]:
                dx = torch.empty_like(x)
                _lib.check(lib.exa_ssim_backward(B * C, H, W, _ptr(x), _ptr(y), _ptr(g), _ptr(a0), _ptr(a1), _ptr(a2),
                                                 _ptr(dx), _stream_ptr(x.device)))
            if ctx.need[1]:
                dy = torch.empty_like(x)
                _lib.check(lib.exa_ssim_backward(B * C, H, W, _ptr(y), _ptr(x), _ptr(g), _ptr(b0), _ptr(b1), _ptr(b2),
                                                 _ptr(dy), _stream_ptr(x.device)))
        return dx, dy


class SSIM(nn.Module):
    """``forward(img_out, img_target, bbox=None, mask=None, window_size=11)`` -> SSIM map, as reference loss.py:31-74."""

    def __init__(self):
        super(SSIM, self).__init__()

    def forward(self, img_out, img_target, bbox=None, mask=None, window_size=11):
        if window_size != 11:
            raise NotImplementedError('exavatar_release_amd: the fused SSIM implements the reference\'s window_size = 11')
        if mask is not None:
            img_out, img_target = img_out * mask, img_target * mask
        if bbox is not None:
            x0, y0, w, h = _crop_window(bbox, img_out.shape[2], img_out.shape[3])
            img_out, img_target = img_out[:, :, y0:y0 + h, x0:x0 + w], img_target[:, :, y0:y0 + h, x0:x0 + w]
        return _FusedSSIM.apply(img_out, img_target)


class _Photometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img_out, img_target, l1_weight, ssim_mask, crop, w_l1, w_ssim):
        lib = _lib.load()
        device = img_out.device
        _need_rocm(device, 'PhotometricLoss')
        if img_out.dim() != 4 or img_out.shape != img_target.shape:
            raise ValueError('PhotometricLoss expects two [B, C, H, W] images of the same shape')
        x, y = _dev_f32(img_out, device, 'img_out'), _dev_f32(img_target, device, 'img_target')
        B, C, H, W = x.shape
        lw = _dev_f32(l1_weight, device, 'l1_weight')
        sm = _dev_f32(ssim_mask, device, 'ssim_mask')
        lw = lw.expand(B, 1, H, W).contiguous() if lw is not None else None
        sm = sm.expand(B, 1, H, W).contiguous() if sm is not None else None
        cw, ch = crop[2], crop[3]
        n = B * C * cw * ch
        if n == 0:
            raise ValueError('PhotometricLoss: empty crop window')
        nblk = int(lib.exa_photo_loss_blocks(B, C, cw, ch))
        maps = torch.empty(3 * n, dtype=torch.float32, device=device)
        partials = torch.empty((nblk, 2), dtype=torch.float32, device=device)
        full = crop == (0, 0, W, H)
        dimg = torch.empty_like(x) if full else torch.zeros_like(x)
        with torch.cuda.device(device):
            st, cc = _stream_ptr(device), _c_crop(crop)
            _lib.check(lib.exa_photo_loss_forward(B, C, H, W, cc, _ptr(x), _ptr(y), _ptr(lw), _ptr(sm), _ptr(maps),
                                                  _ptr(partials), st))
            need = ctx.needs_input_grad[0]
            if need:
                _lib.check(lib.exa_photo_loss_grad(B, C, H, W, cc, _ptr(x), _ptr(y), _ptr(lw), _ptr(sm), float(w_l1),
                                                   float(w_ssim), _ptr(maps), _ptr(dimg), st))
        sums = partials.sum(0)                                   # [sum ssim, sum l1]
        l1_mean, ssim_mean = sums[1] / n, sums[0] / n
        loss = w_l1 * l1_mean + w_ssim * (1.0 - ssim_mean)
        if need:
            ctx.save_for_backward(dimg)
        ctx.mark_non_differentiable(l1_mean, ssim_mean)
        return loss, l1_mean, ssim_mean

    @staticmethod
    def backward(ctx, g_loss, _g1, _g2):
        (dimg,) = ctx.saved_tensors
        return dimg * g_loss, None, None, None, None, None, None


class PhotometricLoss(nn.Module):
    """Fused ``w_l1 * mean(l1_weight * |x - y|) + w_ssim * mean(1 - ssim(x * ssim_mask, y * ssim_mask))`` over the
    bbox crop; defaults = the reference's weights (``avatar/main/config.py:35-36``).

    Equals, for the reference's human renders (model.py:197-198),
    ``(rgb_loss(x, y, bbox=bbox) * 0.8).mean() + ((1 - ssim(x, y, bbox=bbox)) * 0.2).mean()`` and, for its scene render
    (model.py:214-215) with ``l1_weight = ssim_mask = 1 - mask``,
    ``(rgb_loss(x, y) * (1 - mask) * 0.8).mean() + ((1 - ssim(x, y, mask=1 - mask)) * 0.2).mean()``.
    ``forward`` returns the scalar loss; ``return_terms=True`` also returns the (detached) means of the L1 map and of
    the SSIM map.  The gradient w.r.t. ``img_out`` is produced by the forward call itself (second kernel) and only
    scaled by the incoming gradient in ``backward``; ``img_target`` gets no gradient."""

    def __init__(self, rgb_loss_weight=0.8, ssim_loss_weight=0.2):
        super(PhotometricLoss, self).__init__()
        self.rgb_loss_weight = float(rgb_loss_weight)
        self.ssim_loss_weight = float(ssim_loss_weight)

    def forward(self, img_out, img_target, bbox=None, l1_weight=None, ssim_mask=None, return_terms=False):
        crop = _crop_window(bbox, img_out.shape[2], img_out.shape[3])
        loss, l1_mean, ssim_mean = _Photometric.apply(img_out, img_target, l1_weight, ssim_mask, crop,
                                                      self.rgb_loss_weight, self.ssim_loss_weight)
        return (loss, l1_mean, ssim_mean) if return_terms else loss
