"""View-sharded data parallelism for the rasterize hot path.

The reference trains with batch size 1 on one GPU (``avatar/main/config.py:44-45``) and loops over
samples in Python (``avatar/main/model.py:81``); every view's rasterize forward/backward is
independent given the shared Gaussian parameters, so views shard across GPUs with exactly one exchange
step: the sum of the parameter gradients.  One process per GPU, ``torch.distributed`` (backend
``nccl`` = RCCL over xGMI on MI355X, ``gloo`` on CPU for the tests).

* :func:`shard_views`         -- round-robin deal of the epoch's (shuffled) views, like
  ``DataLoader(shuffle=True)`` (reference ``avatar/common/base.py:115``).
* :class:`FlatGradAllReducer` -- packs the gradients of the rasterizer inputs into ONE flat fp32 buffer
  (14 floats per Gaussian = 8.4 MB at 150 k: latency-bound on xGMI, so one collective, not five) and
  all-reduces it asynchronously so it overlaps with the next view's rasterize.
* :func:`reduce_densify_stats` -- the densification statistics need their own reductions: SUM of the
  per-view screen-space gradient norms and visibility counts (``module.py:155-157``), MAX of the radii
  (``model.py:284``).
"""
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_views(n_views: int, rank: int, world_size: int, epoch: int = 0, shuffle: bool = True,
                seed: int = 0) -> List[int]:
    """Views of this rank for ``epoch``: a seed-synchronised permutation dealt round-robin.

    Every rank computes the same permutation (seed + epoch), so the shards are disjoint and cover all
    views; ranks get ``ceil`` / ``floor`` shares when ``n_views % world_size != 0``.
    """
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        order = torch.randperm(n_views, generator=g).tolist()
    else:
        order = list(range(n_views))
    return order[rank::world_size]


class FlatGradAllReducer:
    """All-reduce a fixed set of gradient tensors through one persistent flat buffer."""

    def __init__(self, like: Sequence[torch.Tensor], average: bool = True, group=None):
        self.shapes = [t.shape for t in like]
        self.numels = [t.numel() for t in like]
        dev = like[0].device
        self.flat = torch.zeros(sum(self.numels), dtype=torch.float32, device=dev)
        self.views = []
        o = 0
        for n, s in zip(self.numels, self.shapes):
            self.views.append(self.flat[o:o + n].view(s))
            o += n
        self.average = average
        self.group = group
        self._work = None

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    def start(self, grads: Sequence[Optional[torch.Tensor]]):
        """Pack ``grads`` (``None`` = zeros, e.g. a rank that did not touch a parameter) and launch the
        asynchronous all-reduce.  Call :meth:`finish` before reading :attr:`views`."""
        self.finish()
        for v, g in zip(self.views, grads):
            if g is None:
                v.zero_()
            else:
                v.copy_(g)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return self

    def finish(self) -> List[torch.Tensor]:
        if self._work is not None:
            self._work.wait()
            self._work = None
            if self.average:
                self.flat.div_(dist.get_world_size(self.group))
        return self.views


def reduce_densify_stats(grad_norm_accum: torch.Tensor, track_cnt: torch.Tensor, radius_max: torch.Tensor,
                         group=None):
    """In-place cross-rank reduction of the densification statistics: SUM, SUM, MAX."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return grad_norm_accum, track_cnt, radius_max
    dist.all_reduce(grad_norm_accum, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(track_cnt, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(radius_max, op=dist.ReduceOp.MAX, group=group)
    return grad_norm_accum, track_cnt, radius_max
