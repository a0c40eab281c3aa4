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


# ---- clone / split / prune of a Gaussian set, replicated-safe --------------------------------------------------------
def synchronised_generator(seed, device):
    """A ``torch.Generator`` on ``device`` seeded with ``seed``: created with the same seed on every rank of a
    data-parallel run (and advanced in lockstep: the replicas densify from the same REDUCED statistics), it makes the
    random samples of :func:`densify_and_prune` -- the reference draws them from the global CUDA RNG,
    ``torch.normal`` at ``avatar/common/nets/module.py:198`` -- identical on all ranks, so the topology of the replicated
    Gaussian set never diverges."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def _quat_to_matrix(q):
    q = q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)
    w, x, y, z = q.unbind(1)
    return torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)), 1).view(-1, 3, 3)


def densify_and_prune(params, optimizer, grad_accum, track_cnt, *, grad_thr, extent, dense_percent=0.01, opacity_min=0.005,
                      prune_big=False, split_factor=2, generator=None, rotation_to_matrix=None, radius_max=None,
                      screen_size_max=None, reference_radius_reset=True):
    """Clone / split / prune one set of Gaussians and carry the optimizer state along -- what the reference's
    ``SceneGaussian.densify_and_prune`` does with its Adam-state surgery (``avatar/common/nets/module.py:17-72,159-240``),
    restated for a plain dict of parameters and any ``torch.optim`` optimizer with per-parameter state tensors.

    ``params``: dict name -> ``nn.Parameter`` with first dimension P; needs ``'mean'`` [P, 3], ``'scale'`` [P, 3] (LOG
    scales, activation ``exp``), ``'rotation'`` ([P, 4] quaternion w, x, y, z -- or give ``rotation_to_matrix``) and
    ``'opacity'`` [P, 1] (logits, activation ``sigmoid``); every other entry (colours, SH) is carried along.  Each
    parameter must be the single member of an optimizer group.  ``grad_accum`` / ``track_cnt``: the statistics of
    :func:`track_densify_stats` (already reduced over the ranks: ``dist.reduce_densify_stats``).

    Order of events as in the reference: mean screen-space gradient ``>= grad_thr`` selects; small ones (max scale
    ``<= dense_percent * extent``) are CLONED, large ones SPLIT into ``split_factor`` samples of their own Gaussian (scale
    divided by ``0.8 * split_factor``) and removed; then everything with opacity ``< opacity_min`` (and, with
    ``prune_big``, max scale ``> 0.1 * extent``) is pruned.  Row order of the result: surviving originals, clones,
    split samples.  New rows start with zero optimizer state.  ``generator``: see :func:`synchronised_generator`.

    Screen-space prune (``prune_big`` with ``radius_max`` [P] and ``screen_size_max``; the reference passes 20 px once
    ``cur_itr > opacity_reset_interval``, ``avatar/main/model.py:288-289``): the reference tests ``self.radius_max >
    screen_size_max`` AFTER ``clone_points`` / ``split_points``, whose ``densify()`` has just replaced ``radius_max`` by zeros
    for EVERY row (module.py:223) -- the test can never fire there.  ``reference_radius_reset=True`` (default) reproduces
    that (topology identical to the reference's); ``False`` applies the criterion the code evidently intends (upstream
    3DGS): surviving originals keep their ``radius_max``, new rows start at zero.

    Returns ``(new_params, n_cloned, n_split, n_pruned)``; the caller allocates fresh statistics of the new length (the
    reference zeroes them in ``densify``, module.py:223-225)."""
    mean, scale, rot, opac = params['mean'], params['scale'], params['rotation'], params['opacity']
    P0 = mean.shape[0]
    with torch.no_grad():
        g = torch.nan_to_num(grad_accum.reshape(-1) / track_cnt.reshape(-1))
        big = torch.exp(scale).max(1).values > dense_percent * extent
        hot = g >= grad_thr
        clone_rows = torch.nonzero(hot & ~big).flatten()
        split_rows = torch.nonzero(hot & big).flatten()
        keep_rows = torch.nonzero(~(hot & big)).flatten()
        rep = split_rows.repeat(split_factor)
        std = torch.exp(scale[rep])
        noise = torch.randn(std.shape, generator=generator, device=std.device, dtype=std.dtype) * std
        R = (rotation_to_matrix or _quat_to_matrix)(rot[rep])
        new = {}
        for name, p in params.items():
            parts = [p[keep_rows], p[clone_rows]]
            if name == 'mean':
                parts.append(torch.bmm(R, noise[:, :, None])[:, :, 0] + p[rep])
            elif name == 'scale':
                parts.append(torch.log(torch.exp(p[rep]) / (0.8 * split_factor)))
            else:
                parts.append(p[rep])
            new[name] = torch.cat(parts)
        # rows of the ORIGINAL set each new row descends from with its optimizer state (-1: a new row, zero state)
        src = torch.cat((keep_rows, torch.full((clone_rows.numel() + rep.numel(),), -1, dtype=torch.long, device=keep_rows.device)))
        drop = torch.sigmoid(new['opacity'])[:, 0] < opacity_min
        if prune_big:
            drop |= torch.exp(new['scale']).max(1).values > 0.1 * extent
            if radius_max is not None and screen_size_max and not reference_radius_reset:
                r_new = torch.zeros(src.numel(), dtype=torch.float32, device=src.device)
                r_new[:keep_rows.numel()] = radius_max.reshape(-1).to(torch.float32)[keep_rows]
                drop |= r_new > float(screen_size_max)
        valid = ~drop
        src = src[valid]
        out = {}
        for group in optimizer.param_groups:
            if len(group['params']) != 1:
                continue
            old = group['params'][0]
            name = next((n for n, p in params.items() if p is old), None)
            if name is None:
                continue
            fresh = torch.nn.Parameter(new[name][valid].contiguous().requires_grad_(True))
            state = optimizer.state.pop(old, None)
            if state is not None:
                for k, v in list(state.items()):
                    if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == P0:
                        nv = torch.zeros((src.numel(),) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
                        old_rows = src >= 0
                        nv[old_rows] = v[src[old_rows]]
                        state[k] = nv
                optimizer.state[fresh] = state
            group['params'][0] = fresh
            out[name] = fresh
        missing = [n for n in params if n not in out]
        if missing:
            raise ValueError('densify_and_prune: parameters %s are not single members of an optimizer group' % missing)
    return out, int(clone_rows.numel()), int(split_rows.numel()), int(drop.sum())
