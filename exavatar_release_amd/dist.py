"""View-sharded data parallelism for the rasterize hot path.

The reference trains with batch size 1 on one GPU (``avatar/main/config.py:44-45``) and loops over
samples in Python (``avatar/main/model.py:81``); every view's rasterize forward/backward is
independent given the shared Gaussian parameters, so views shard across GPUs with exactly one exchange
step: the sum of the parameter gradients.  One process per GPU, ``torch.distributed`` (backend
``nccl`` = RCCL over xGMI on MI355X, ``gloo`` on CPU for the tests).

* :func:`shard_views`         -- round-robin deal of the epoch's (shuffled) views, like
  ``DataLoader(shuffle=True)`` (reference ``avatar/common/base.py:115``), equal shard lengths on all ranks.
* :class:`FlatGradAllReducer` -- packs the gradients of the rasterizer inputs into ONE flat fp32 buffer
  (14 floats per Gaussian = 8.4 MB at 150 k: latency-bound on xGMI, so one collective, not five) and
  all-reduces it asynchronously so it overlaps with the next view's rasterize; double-buffered and
  hipGraph-safe (``bench.py --gpus N`` runs on exactly this class).
* :func:`reduce_densify_stats` -- the densification statistics need their own reductions: SUM of the
  per-view screen-space gradient norms and visibility counts (``module.py:155-157``), MAX of the radii
  (``model.py:284``).
"""
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_views(n_views: int, rank: int, world_size: int, epoch: int = 0, shuffle: bool = True,
                seed: int = 0, pad: bool = True, order: Optional[Sequence[int]] = None) -> List[int]:
    """Views of this rank for ``epoch``: a seed-synchronised permutation dealt round-robin.

    ``order`` (a permutation of ``range(n_views)``, the same on every rank) replaces the seeded shuffle, e.g. a
    stratified order in which every short window of steps covers the camera ring evenly (``bench.py``).

    Every rank computes the same permutation (seed + epoch), so the shards cover all views.  With ``pad`` (default)
    every rank gets exactly ``ceil(n_views / world_size)`` views -- the permutation is extended by wrapping around,
    like ``DistributedSampler`` does -- so that all ranks issue the same number of gradient all-reduces per epoch
    (ranks with a shorter shard would otherwise leave the others hanging in their last collective).  ``pad=False``
    deals the ``ceil`` / ``floor`` shares of a plain partition; the caller must then call
    :meth:`FlatGradAllReducer.start` with ``None`` gradients for the missing steps.
    """
    if order is not None:
        order = [int(v) for v in order]
        if sorted(order) != list(range(n_views)):
            raise ValueError('shard_views: order must be a permutation of range(n_views)')
    elif shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        order = torch.randperm(n_views, generator=g).tolist()
    else:
        order = list(range(n_views))
    if pad and n_views > 0 and n_views % world_size:
        total = -(-n_views // world_size) * world_size            # repeat the permutation as DistributedSampler does, so
        order = (order * (-(-total // n_views)))[:total]          # that even n_views < world_size leaves no rank empty
    return order[rank::world_size]


class FlatGradAllReducer:
    """All-reduce a fixed set of gradient tensors through persistent flat fp32 buffers.

    * ONE collective per step (14 floats per Gaussian = 8.4 MB at 150 k: latency-bound on xGMI, so one all-reduce,
      not five).
    * ``n_buffers = 2`` double-buffers: the all-reduce of step i reads buffer i % 2 while step i + 1 packs into the
      other one, so the collective overlaps the next view's rasterize without a copy being overwritten under RCCL.
    * hipGraph-safe packing: :meth:`pack` uses elementwise kernels (``torch.mul(g, 1, out=view)``), never ``copy_``
      (a D2D ``copy_`` becomes a memcpy graph node, which breaks stream capture in the ROCm runtime bundled with
      torch 2.10); capture ``pack`` together with the rasterize step, then call :meth:`reduce` after each replay.
    """

    def __init__(self, like: Sequence[torch.Tensor], average: bool = True, group=None, n_buffers: int = 1):
        self.shapes = [t.shape for t in like]
        self.numels = [t.numel() for t in like]
        dev = like[0].device
        self.n_buffers = max(1, int(n_buffers))
        self.flats = [torch.zeros(sum(self.numels), dtype=torch.float32, device=dev) for _ in range(self.n_buffers)]
        self._views = []
        for flat in self.flats:
            vs, o = [], 0
            for n, s in zip(self.numels, self.shapes):
                vs.append(flat[o:o + n].view(s))
                o += n
            self._views.append(vs)
        self.average = average
        self.group = group
        self._work = [None] * self.n_buffers
        self._next = 0
        self._last = 0

    # single-buffer aliases kept for callers that read them directly
    @property
    def flat(self) -> torch.Tensor:
        return self.flats[self._last]

    @property
    def views(self) -> List[torch.Tensor]:
        return self._views[self._last]

    @property
    def nbytes(self) -> int:
        return self.flats[0].numel() * 4

    def buffer_views(self, b: int) -> List[torch.Tensor]:
        return self._views[b]

    def _multi(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def wait(self, b: int):
        """Wait for the collective that last used buffer ``b`` (and apply the averaging)."""
        if self._work[b] is not None:
            self._work[b].wait()
            self._work[b] = None
            if self.average:
                self.flats[b].div_(dist.get_world_size(self.group))

    def pack(self, grads: Sequence[Optional[torch.Tensor]], b: int = 0):
        """Write ``grads`` (``None`` = zeros, e.g. a rank that did not touch a parameter) into buffer ``b`` with
        elementwise kernels only (capturable in a hipGraph)."""
        for v, g in zip(self._views[b], grads):
            if g is None:
                v.zero_()
            else:
                torch.mul(g.view_as(v) if g.shape != v.shape else g, 1.0, out=v)

    def reduce(self, b: int = 0):
        """Launch the asynchronous all-reduce of buffer ``b`` (no-op for a single rank)."""
        self._last = b
        if self._multi():
            self._work[b] = dist.all_reduce(self.flats[b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return self

    def start(self, grads: Sequence[Optional[torch.Tensor]]):
        """Pack ``grads`` into the next buffer and launch its asynchronous all-reduce.  With two buffers the previous
        step's collective keeps running; call :meth:`finish` before reading :attr:`views`."""
        b = self._next
        self._next = (b + 1) % self.n_buffers
        self.wait(b)
        self.pack(grads, b)
        return self.reduce(b)

    def finish(self) -> List[torch.Tensor]:
        """Wait for every outstanding collective; returns the views of the most recently reduced buffer."""
        for b in range(self.n_buffers):
            self.wait(b)
        return self._views[self._last]


def reduce_densify_stats(grad_norm_accum: torch.Tensor, track_cnt: torch.Tensor, radius_max: torch.Tensor,
                         group=None):
    """In-place cross-rank reduction of the densification statistics: SUM, SUM, MAX."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return grad_norm_accum, track_cnt, radius_max
    dist.all_reduce(grad_norm_accum, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(track_cnt, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(radius_max, op=dist.ReduceOp.MAX, group=group)
    return grad_norm_accum, track_cnt, radius_max
