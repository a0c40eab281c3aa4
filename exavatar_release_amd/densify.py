"""Fused densification statistics.

Host-side mirror of what the reference does with the rasterizer's outputs after every backward on the scene
Gaussians: ``avatar/main/train.py:49-54`` stacks ``mean_2d.grad``, ``avatar/main/model.py:279-285`` keeps the
per-Gaussian maximum screen radius, ``avatar/common/nets/module.py:155-157`` (``SceneGaussian.track_stats``)
accumulates ``||grad[:, :2]||`` and a visit count -- three boolean-mask statements per render, each of which
synchronises the host (``nonzero``).  Here it is one streaming HIP kernel on the tensors' own stream, no sync;
in a multi-GPU run the statistics are then combined with ``dist.reduce_densify_stats``.
"""
import torch

from . import _lib
from .rasterizer import _ptr, _stream_ptr


def track_densify_stats(mean_2d_grad, radius, xyz_grad_accum=None, track_cnt=None, radius_max=None):
    """In place, for every Gaussian with ``radius > 0`` (the reference's ``is_vis``):
    ``xyz_grad_accum += ||mean_2d_grad[:, :2]||``, ``track_cnt += 1``, ``radius_max = max(radius_max, radius)``.

    ``mean_2d_grad``: float32 [P, 3] (``mean_2d.grad`` of one render); ``radius``: int32 [P] as returned by the
    rasterizer; the three statistics: float32 with P elements ([P] or [P, 1]), any of them may be ``None``.
    """
    lib = _lib.load()
    device = radius.device
    if device.type != 'cuda':
        raise RuntimeError('exavatar_release_amd: track_densify_stats runs on a ROCm device only (no CPU path)')
    P = int(radius.shape[0])
    if radius.dtype != torch.int32 or not radius.is_contiguous():
        raise ValueError('radius must be a contiguous int32 tensor (the rasterizer\'s `radii` output)')
    outs = []
    for name, t in (('xyz_grad_accum', xyz_grad_accum), ('track_cnt', track_cnt), ('radius_max', radius_max)):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != P or t.device != device):
            raise ValueError('%s must be a contiguous float32 tensor with %d elements on %s' % (name, P, device))
        outs.append(t)
    g = None
    if xyz_grad_accum is not None:
        if mean_2d_grad is None or tuple(mean_2d_grad.shape) != (P, 3):
            raise ValueError('mean_2d_grad must have shape [%d, 3]' % P)
        g = mean_2d_grad.detach().to(device=device, dtype=torch.float32).contiguous()
    with torch.no_grad(), torch.cuda.device(device):
        _lib.check(lib.exa_raster_densify_stats(P, _ptr(g), _ptr(radius), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]),
                                                _stream_ptr(device)))
