"""Image losses in front of the rasterizer's backward.

Host-side mirrors of the two image losses of the reference that produce ``dL/d(render)``:

* ``RGBLoss``  -- reference ``avatar/common/nets/loss.py:11-29`` (L1 map with optional mask / background / bbox):
  plain elementwise tensor ops, kept as such.
* ``SSIM``     -- reference ``avatar/common/nets/loss.py:31-74``: the map itself runs as ONE fused HIP kernel
  (``csrc/ssim.hip``) instead of five grouped 11x11 ``conv2d`` calls and ~15 elementwise kernels, and its backward
  as one more; mask multiplication and bbox cropping stay ordinary (differentiable) tensor ops in front of it.

Same constructor / ``forward`` signatures, argument meaning and results; ROCm device tensors only (no CPU path).
"""
import torch
import torch.nn as nn

from . import _lib
from .rasterizer import _ptr, _stream_ptr


class RGBLoss(nn.Module):
    def __init__(self):
        super(RGBLoss, self).__init__()

    def forward(self, img_out, img_target, bbox=None, mask=None, bg=None):
        if (mask is not None) and (bg is not None):
            img_target = img_target * mask + (1 - mask) * bg[:, :, None, None]
        if bbox is not None:
            img_height, img_width = img_out.shape[2:]
            xmin, ymin, width, height = [int(x) for x in bbox[0]]
            xmin = max(xmin, 0)
            ymin = max(ymin, 0)
            xmax = min(xmin + width, img_width)
            ymax = min(ymin + height, img_height)
            img_out = img_out[:, :, ymin:ymax, xmin:xmax]
            img_target = img_target[:, :, ymin:ymax, xmin:xmax]
        return torch.abs(img_out - img_target)


class _FusedSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img_out, img_target):
        lib = _lib.load()
        device = img_out.device
        if device.type != 'cuda':
            raise RuntimeError('exavatar_release_amd: the fused SSIM runs on a ROCm device only (no CPU path)')
        x = img_out.detach().to(torch.float32).contiguous()
        y = img_target.detach().to(device=device, dtype=torch.float32).contiguous()
        if x.dim() != 4 or x.shape != y.shape:
            raise ValueError('SSIM expects two [B, C, H, W] images of the same shape')
        B, C, H, W = x.shape
        need_grad = ctx.needs_input_grad[0]
        if ctx.needs_input_grad[1]:
            raise NotImplementedError('exavatar_release_amd: SSIM gradient w.r.t. the target image is not implemented')
        out = torch.empty_like(x)
        maps = [torch.empty_like(x) for _ in range(3)] if need_grad else [None, None, None]
        with torch.cuda.device(device):
            _lib.check(lib.exa_ssim_forward(B * C, H, W, _ptr(x), _ptr(y), _ptr(out), _ptr(maps[0]), _ptr(maps[1]),
                                            _ptr(maps[2]), _stream_ptr(device)))
        if need_grad:
            ctx.save_for_backward(x, y, *maps)
        return out

    @staticmethod
    def backward(ctx, grad_map):
        lib = _lib.load()
        x, y, m0, m1, m2 = ctx.saved_tensors
        B, C, H, W = x.shape
        g = grad_map.to(torch.float32).expand(x.shape).contiguous()
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.exa_ssim_backward(B * C, H, W, _ptr(x), _ptr(y), _ptr(g), _ptr(m0), _ptr(m1), _ptr(m2), _ptr(dx),
                                             _stream_ptr(x.device)))
        return dx, None


class SSIM(nn.Module):
    def __init__(self):
        super(SSIM, self).__init__()

    def forward(self, img_out, img_target, bbox=None, mask=None, window_size=11):
        if window_size != 11:
            raise NotImplementedError('exavatar_release_amd: the fused SSIM implements the reference\'s window_size = 11')
        batch_size, feat_dim, img_height, img_width = img_out.shape
        if mask is not None:
            img_out = img_out * mask
            img_target = img_target * mask
        if bbox is not None:
            xmin, ymin, width, height = [int(x) for x in bbox[0]]
            xmin = max(xmin, 0)
            ymin = max(ymin, 0)
            xmax = min(xmin + width, img_width)
            ymax = min(ymin + height, img_height)
            img_out = img_out[:, :, ymin:ymax, xmin:xmax]
            img_target = img_target[:, :, ymin:ymax, xmin:xmax]
        return _FusedSSIM.apply(img_out, img_target)
