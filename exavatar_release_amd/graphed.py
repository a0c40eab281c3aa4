"""hipGraph-captured training iteration: the five same-camera renders of one ExAvatar training sample
(reference ``avatar/main/model.py:119-167``) forward AND backward, as a product class.

The reference's training loop (``avatar/main/train.py:41-57``) issues five ``GaussianRenderer`` calls per sample and one
``backward``; through the eager drop-in surface that is ~0.9 ms per iteration with the host as busy as the GPU
(autograd nodes, ~60 launches, allocator calls).  The number of Gaussians only changes when the scene is densified or
pruned (every 100 iterations, ``avatar/main/config.py:17-20``; the human's count never changes), the image size and the
focal length practically never: between two such events every iteration launches the SAME kernels on the SAME buffers.
:class:`GraphedIteration` captures them once -- one hipGraph for the five forwards, one for their backwards -- and
replays them per iteration::

    it = GraphedIteration((H, W), device)
    for data in loader:
        out = it(scene_asset, human_asset, human_asset_refined, cam_param, bg, scene_densify_stats)   # dict like render_iteration's
        loss = my_losses(out, data)              # any PyTorch code: L1 / SSIM / LPIPS / PhotometricLoss ...
        loss.backward()                          # gradients reach the asset tensors (and out[...]['mean_2d'].grad)
        optimizer.step()

What a call does: copy the asset tensors into the graph's static inputs (one fused multi-tensor copy), write the camera
block with one kernel from the device-resident ``R`` / ``t`` (``exa_raster_camera_block``), replay the forward graph.
``backward`` copies the incoming image gradients into static buffers, replays the backward graph and hands out one
clone of the flat gradient buffer.  The asset tensors may be leaves or the outputs of networks (the reference's are:
``SceneGaussian.forward`` / ``HumanGaussian.forward``); autograd continues into whatever produced them.

Re-capture happens when P of a set changes (densify / prune), the colour input changes (rgb <-> sh), tan(fov) changes,
a render needs more tile instances than its buffer holds (the plain renders report ``{needed, overflow}`` into reserved
pinned-host slots ~35 us into the replay; they are polled right after the replay is queued -- the GPU is busy with the
rest of the forward meanwhile -- and an overflowed iteration is re-captured with enough room and rendered again BEFORE
the call returns: the images handed out are always complete, same contract as the eager path's
report polling), or the densification-statistics tensors are replaced.  Results are bit-identical
to eager :func:`renderer.render_iteration` (same kernels, same order; tests/test_gpu_graphed_iteration.py).

The returned images alias the graph's static outputs: valid until the next call (clone what must survive).

**The loss inside the graph** (``loss_fn``).  With the loss in PyTorch between the two graphs the host is on the critical
path again: after the forward replay it launches the loss kernels, runs the autograd engine through them and only then queues
the backward replay (~0.3 ms of Python per iteration, during part of which the GPU waits).  A loss that is a fixed sequence of
tensor operations can be recorded as well::

    def my_loss(out, target, mask):              # out: the dict of the five renders; tensors only, no .item() / host branches
        return photo(out['scene']['img'][None], target, l1_weight=1 - mask) + ...
    it = GraphedIteration((H, W), device, loss_fn=my_loss)
    for data in loader:
        out = it(scene_asset, human_asset, human_asset_refined, cam_param, bg, stats, loss_args=(data['img'], data['mask']))
        out['loss'].backward()                   # the gradients were computed by the SAME replay; this hands them on
        optimizer.step()

Forward, loss and backward are then ONE hipGraph per iteration; ``out['loss'].backward()`` only scales the (already
computed) flat gradient buffer by the incoming gradient and passes its slices to the asset tensors.  ``loss_args`` are copied
into static tensors of the capture per call (one fused copy; a change of their shapes re-captures); the images in ``out`` carry
no autograd history in this mode (the loss is the only differentiable output).  The densification statistics are updated by
the replay itself; an iteration that overflowed restores them from the copy the graph makes first and runs again.
"""
import ctypes
import time
import warnings

import torch

from . import _lib

from . import rasterizer as rz
from .renderer import ITERATION_RENDERS, _sh_degree, camera_block_device, render_iteration

_ASSET_KEYS = ('mean_3d', 'scale', 'rotation', 'opacity')
_IMG_ONLY = (True, False, False) * 5
_SETS = ('scene', 'human', 'human_refined')


def _colour_key(asset):
    return 'rgb' if _sh_degree(asset) is None else 'sh'


class _Captured:
    """Everything one capture owns (replaced as a whole on re-capture)."""
    __slots__ = ('key', 'inputs', 'in_list', 'in_raw', 'probes', 'tan', 'fwd', 'pool', 'outs', 'out_list', 'radii', 'bwd',
                 'slots', 'caps', 'dens_ptrs', 'sizes', 'diff_inputs', 'offsets', 'strides', 'loss', 'loss_args', 'flat',
                 'unused', 'dens', 'dens_backup', 'slots_c')


class _IterFn(torch.autograd.Function):
    """Autograd boundary of a replayed iteration.  apply(owner, cap, *asset tensors [15], *probes [5]) -> 15 image planes
    (img, depthmap, mask of the five renders)."""

    @staticmethod
    def forward(ctx, owner, cap, *tensors):
        ctx.owner, ctx.cap, ctx.serial = owner, cap, owner._serial
        ctx.set_materialize_grads(False)
        # Aliases of the static outputs: no copies (they are overwritten by the next replay).  Detached, because the static
        # tensors carry the autograd graph of the CAPTURED call, which the backward graphs were recorded from and which
        # must not be re-parented onto this node.
        return tuple(o.detach() for o in cap.out_list)

    @staticmethod
    def backward(ctx, *grads):
        owner = ctx.owner
        cap = owner._cap
        if cap is not ctx.cap:
            raise RuntimeError('exavatar_release_amd: GraphedIteration.backward after a later call replaced its capture '
                               '(call backward before the next iteration)')
        if owner._serial != ctx.serial:
            raise RuntimeError('exavatar_release_amd: GraphedIteration.backward after a later call overwrote the forward '
                               'state its backward graph reads (call backward before the next iteration)')
        flat = owner._backward(grads[:15])
        return (None, None) + _grad_views(flat, cap, ctx.needs_input_grad[2:], None)


def _grad_views(flat, cap, needed, unused):
    """The slices of a flat gradient buffer as tensors of the inputs' shapes (as_strided: a third of the host time of
    slice + view, twenty times per iteration)."""
    outs = []
    for i, need in enumerate(needed):
        if need and not (unused is not None and unused[i]):
            outs.append(torch.as_strided(flat, cap.diff_inputs[i], cap.strides[i], cap.offsets[i]))
        else:
            outs.append(None)
    return tuple(outs)


class _LossFn(torch.autograd.Function):
    """Autograd boundary of an iteration whose loss and backward were part of the replay (``loss_fn``): apply(owner, cap,
    *asset tensors [15], *probes [5]) -> the loss; backward scales the gradients the replay left in ``cap.flat``."""

    @staticmethod
    def forward(ctx, owner, cap, *tensors):
        ctx.owner, ctx.cap, ctx.serial = owner, cap, owner._serial
        return cap.loss.detach()

    @staticmethod
    def backward(ctx, g):
        owner = ctx.owner
        cap = owner._cap
        if cap is not ctx.cap:
            raise RuntimeError('exavatar_release_amd: GraphedIteration: backward of a loss after a later call replaced its '
                               'capture (call backward before the next iteration)')
        if owner._serial != ctx.serial:
            raise RuntimeError('exavatar_release_amd: GraphedIteration: backward of a loss after a later call overwrote its '
                               'gradients (call backward before the next iteration)')
        with rz._on_device(owner.device):
            flat = cap.flat * g                 # a private copy, scaled by dL/dloss (ones for a plain loss.backward())
        return (None, None) + _grad_views(flat, cap, ctx.needs_input_grad[2:], cap.unused)


class GraphedIteration:
    """See the module docstring.  ``merge``: composites as list merges (default) or as constant-prefix renders, as in
    :func:`renderer.render_iteration`.  ``capacity_growth``: head-room of the instance buffers over what the first
    iteration of a capture needed.  ``check``: poll the overflow reports after every forward replay; switch off only when
    the capacities are known to be sufficient.  ``capacities``: instance capacities
    of the three plain renders (scene, human, refined human) for the FIRST capture instead of measuring them with an eager
    iteration.  ``loss_fn``: ``loss_fn(out, *loss_args) -> scalar tensor`` recorded into the graph together with its backward
    (module docstring); calls then take ``loss_args=(tensors...)`` and return the loss as ``out['loss']``.
    Counters: ``captures``, ``overflow_retries``."""

    def __init__(self, img_shape, device, merge=True, capacity_growth=1.5, check=True, capacities=None, loss_fn=None):
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError('exavatar_release_amd: GraphedIteration runs on a ROCm device only')
        self.shape, self.device = (int(img_shape[0]), int(img_shape[1])), device
        self.merge, self.growth, self.check = bool(merge), float(capacity_growth), bool(check)
        self.loss_fn = loss_fn
        self._pool, self._keeper = None, None            # ONE memory pool for every recording of this object (see _release)
        self.tight_backward = True      # developer A/B knob: bake the batch slots in use into the backward launches (_backward)
        self._serial = 0                # number of forward replays: a backward must belong to the latest one
        self._cam = torch.zeros(38, dtype=torch.float32, device=device)   # viewmatrix 16 | projmatrix 16 | campos 3 | bg 3:
        #                                                                   outlives the captures, which read views of it
        # pointer table of the backward graphs (ExaRasterBackwardJob.dL_dcolor_indirect): entry i holds the address the
        # backward of render i reads dL/dimg from -- autograd's own tensor of this iteration, no copy into a static buffer
        self._ptr_table = torch.zeros(16, dtype=torch.int64, device=device)
        self._cap = None
        self._caps_hint = None if capacities is None else [int(c) for c in capacities]   # capacities of the next capture
        self._intr, self._focal_src, self._focal_ver = None, None, None
        self._bg_src, self._bg_ver = None, None
        self._last_needs = None         # (P_scene, P_human, needs of the three plain renders) of the last checked iteration
        self._reports_checked = True
        self.captures = 0
        self.backward_captures = 0
        self.overflow_retries = 0

    # ---- capture ---------------------------------------------------------------------------------------------------
    def _release(self):
        cap, self._cap = self._cap, None
        if cap is None:
            return
        # A graph must not be destroyed while one of its replays is still executing (the runtime frees its kernel-argument
        # and node storage with it): wait for the device first.  Re-captures and close() are rare (a change of P, an
        # overflow, the end of training), the wait costs them nothing.
        torch.cuda.synchronize(self.device)
        if rz._hdr_pool is not None:
            for s in (cap.slots or []) + (getattr(cap, 'slots_c', None) or []):
                if s is not None:
                    rz._hdr_pool.release(s[0])
        # Let go of the recordings and their static tensors NOW (not whenever the cyclic collector finds the old capture): their
        # memory then goes back to this object's graph pool and the next recording re-uses it.  A pool of its own per capture
        # made every change of P pay ~90 ms of hipFree for the previous capture's segments (tools/gpu_capture_cost.py).
        for name in ('fwd', 'bwd', 'outs', 'out_list', 'radii', 'inputs', 'in_list', 'in_raw', 'probes', 'loss', 'loss_args', 'flat',
                     'dens_backup'):
            try:
                setattr(cap, name, None)
            except AttributeError:
                pass

    def close(self):
        """Release the captured graphs, their static tensors and the reserved report slots (after waiting for the device).
        The object can be used again afterwards (it captures anew).  Call it when a training run ends; ``with
        GraphedIteration(...) as it:`` does."""
        self._release()
        self._pool, self._keeper = None, None          # (the pool dies with its last graph; the next capture starts a new one)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            if self._cap is not None:
                warnings.warn('exavatar_release_amd: GraphedIteration was garbage-collected with a live capture; call close()',
                              ResourceWarning, stacklevel=2)
                if torch.cuda.is_current_stream_capturing():
                    # collected in the middle of SOMEBODY ELSE's stream capture: the device wait of close() is illegal there and
                    # would invalidate that capture.  Give the report slots back and let the graphs go with the object; the
                    # replays they belong to were queued before the capture began.
                    cap, self._cap = self._cap, None
                    if rz._hdr_pool is not None:
                        for s in (cap.slots or []) + (getattr(cap, 'slots_c', None) or []):
                            if s is not None:
                                rz._hdr_pool.release(s[0])
                else:
                    self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def _static_like(self, assets):
        """Static input leaves of a capture: slices of ONE flat buffer (filled by one concatenation kernel per call)."""
        f32 = dict(dtype=torch.float32, device=self.device)
        shapes = [(name, k, tuple(a[k].shape)) for name, a in zip(_SETS, assets) for k in _ASSET_KEYS + (_colour_key(a),)]
        numel = [int(torch.Size(sh).numel()) for _, _, sh in shapes]
        pad = [(n + 3) // 4 * 4 for n in numel]                     # 16-byte aligned slices (float4 loads in the kernels)
        flat = torch.zeros(sum(pad), **f32)
        out, off = {name: {} for name in _SETS}, 0
        for (name, k, sh), n, pn in zip(shapes, numel, pad):
            out[name][k] = flat[off:off + n].view(sh).requires_grad_(True)
            off += pn
        for name, a in zip(_SETS, assets):
            if _colour_key(a) == 'sh' and a.get('sh_degree') is not None:
                out[name]['sh_degree'] = int(a['sh_degree'])
        self._flat_in, self._flat_in_tight = flat.data, numel == pad     # (.data: its own version counter, see in_raw)
        return out

    def _render(self, cap, dens):
        c = self._cam
        block = (cap.tan[0], cap.tan[1], c[0:16].view(4, 4), c[16:32].view(4, 4), c[32:35])
        i = cap.inputs
        return render_iteration(None, i['scene'], i['human'], i['human_refined'], self.shape, None, c[35:38], dens,
                                merge=self.merge, cam_block=block, probes=cap.probes)

    def _capture(self, assets, tan, dens, key, loss_args=()):
        dev = self.device
        self._release()
        cap = _Captured()
        cap.key, cap.tan, cap.slots, cap.slots_c = key, tan, [], []
        cap.inputs = self._static_like(assets)
        cap.in_list = [cap.inputs[n][k] for n, a in zip(_SETS, assets) for k in _ASSET_KEYS + (_colour_key(a),)]
        # The per-iteration values are written through `.data` aliases (same storage, their own version counters): the
        # captured autograd graph saved the static leaves for its backward, and a copy_ that bumped THEIR counters would
        # make a later recording of another backward pattern fail the saved-tensor check although nothing is stale (the
        # kernels read the buffers at replay time).
        cap.in_raw = [t.data for t in cap.in_list]
        Ps, Ph = cap.inputs['scene']['mean_3d'].shape[0], cap.inputs['human']['mean_3d'].shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        cap.probes = [torch.zeros((n, 3), **f32).requires_grad_(True) for n in (Ps, Ph, Ph, Ph, Ph)]
        cap.dens_ptrs = None if dens is None else tuple(None if t is None else t.data_ptr() for t in dens)
        cap.diff_inputs = [tuple(t.shape) for t in cap.in_list + cap.probes]
        cap.sizes = [t.numel() for t in cap.in_list + cap.probes]
        cap.strides = [tuple(t.stride()) for t in cap.in_list + cap.probes]
        cap.offsets = [sum(cap.sizes[:i]) for i in range(len(cap.sizes))]
        cap.loss = cap.flat = cap.unused = cap.dens_backup = None
        cap.dens = None if dens is None else [t for t in dens if t is not None]
        cap.loss_args = [t.detach().clone() for t in loss_args]          # static: the recorded loss reads these
        self._fill_inputs(cap, assets)
        saved = (rz.config.mode, rz.config.fixed_capacity, rz.config.on_overflow)
        # (the eager renders below run with capacities that may be estimates: an overflow there is repaired in place whatever the
        #  user's policy says -- the captured graph has its own overflow path, _grow)
        rz.config.on_overflow = 'retry'
        try:
            with torch.enable_grad():
                # Capacities of the three plain renders (scene, human, refined human).  After an overflow: what the reports
                # asked for.  After a change of P (densify / prune): the last measured needs scaled by the change -- a
                # re-capture then costs two recordings, no eager pass, and an estimate that turns out too small is caught
                # by the overflow reports like any other.  Otherwise one eager iteration with the two-stage protocol.
                caps = self._caps_hint
                if caps is None and self._last_needs is not None:
                    ps0, ph0, n0 = self._last_needs
                    rs, rh = Ps / max(ps0, 1), Ph / max(ph0, 1)
                    caps = [int(n0[0] * max(rs, 1.0) * self.growth), int(n0[1] * max(rh, 1.0) * self.growth),
                            int(n0[2] * max(rh, 1.0) * self.growth)]
                if caps is None:
                    rz.config.mode, rz.config.fixed_capacity = 'exact', None
                    self._render(cap, None)
                    torch.cuda.synchronize(dev)
                    H, W = self.shape
                    need = [rz._seen_D.get((dev.index, P, H, W), 0) for P in (Ps, Ph, Ph)]
                    caps = [int(n * self.growth) for n in need]
                caps = [max(c, 64) for c in caps]
                if not self.merge:           # five jobs: scene, human, scene + human, refined, scene + refined
                    caps = [caps[0], caps[1], caps[0] + caps[1], caps[2], caps[0] + caps[2]]
                caps = [(c + 63) // 64 * 64 for c in caps]
                cap.caps = caps
                rz.config.mode, rz.config.fixed_capacity = 'capacity', list(caps)
                if self.captures == 0:
                    # once per object: a warm-up on a side stream (as torch.cuda.graph asks for), forward + backward
                    side = torch.cuda.Stream(device=dev)
                    side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side):
                        res = self._render(cap, None)
                        if self.loss_fn is not None:
                            torch.autograd.grad(self.loss_fn(res, *cap.loss_args), cap.in_list, allow_unused=True)
                        else:
                            torch.autograd.grad([res[k]['img'] for k in ITERATION_RENDERS], cap.in_list,
                                                grad_outputs=[torch.ones_like(res[k]['img']) for k in ITERATION_RENDERS],
                                                allow_unused=True)
                        del res
                    torch.cuda.current_stream(dev).wait_stream(side)
                    torch.cuda.synchronize(dev)
                # header reports of the plain renders go to reserved pinned-host slots (outside the ring eager calls use)
                pool = rz._pool()
                n_jobs = 3 if self.merge else 5
                # (the composites report too -- {slots in use, overflow}, written by the first kernel of their forward: their backward
                #  visits only the batch slots in use, ExaRasterBackwardJob.used_slots, _backward below)
                cap.slots = [None] * n_jobs
                cap.slots_c = [None, None] if self.merge else []
                if pool is not None:
                    taken = []
                    try:                # reserve() raises when the reserved region is exhausted: give back what this capture took
                        for _ in range(n_jobs + len(cap.slots_c)):
                            taken.append(pool.reserve()[:2])
                    except Exception:
                        for slot, _tag in taken:
                            pool.release(slot)
                        raise
                    cap.slots, cap.slots_c = taken[:n_jobs], taken[n_jobs:]
                if self._pool is None:
                    # a pool lives as long as a graph recorded into it: this one-kernel recording keeps it (and the memory
                    # earlier captures gave back to it) across re-captures
                    self._pool = torch.cuda.graph_pool_handle()
                    keeper = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(keeper, pool=self._pool):
                        k_t = torch.zeros(16, device=dev)
                    self._keeper = (keeper, k_t)
                cap.pool = self._pool
                cap.fwd = torch.cuda.CUDAGraph()
                rz._capture_report = cap.slots
                rz._capture_report_c = cap.slots_c
                if self.loss_fn is not None:
                    cap.flat = torch.zeros(sum(cap.sizes), **f32)
                    if cap.dens:
                        cap.dens_backup = [torch.empty_like(t) for t in cap.dens]
                try:
                    with torch.cuda.graph(cap.fwd, pool=cap.pool):
                        if cap.dens_backup:
                            # the replay's backward updates the statistics; an overflowed replay is thrown away and must
                            # leave them as they were (_grow restores them from here)
                            torch._foreach_copy_(cap.dens_backup, cap.dens)
                        res = self._render(cap, dens)
                        if self.loss_fn is not None:
                            loss = self.loss_fn(res, *cap.loss_args)
                            if not (torch.is_tensor(loss) and loss.numel() == 1 and loss.requires_grad):
                                raise ValueError('GraphedIteration: loss_fn must return a scalar tensor that depends on the renders')
                            grads = torch.autograd.grad(loss, cap.in_list + cap.probes, allow_unused=True)
                            cap.unused = self._pack(grads, cap, cap.flat)
                            cap.loss = loss.detach()
                finally:
                    rz._capture_report = rz._capture_report_c = None
                cap.outs = res
                cap.out_list = [res[k][n] for k in ITERATION_RENDERS for n in ('img', 'depthmap', 'mask')]
                # radius / is_vis of the five renders: static tensors the replay rewrites (the composites' are written by their
                # own ranges launch, ExaRasterComposeJob.radii_out)
                cap.radii = {k: (res[k]['radius'], res[k]['is_vis']) for k in ITERATION_RENDERS}
                cap.bwd = {}
                self.captures += 1
        finally:
            rz.config.mode, rz.config.fixed_capacity, rz.config.on_overflow = saved
        # the backward graph of the usual case -- gradients for the five colour images only -- is
        # recorded right away; other patterns (depth / mask gradients, fewer images) on first use
        if self.loss_fn is None and not self.tight_backward:
            self._capture_backward(cap, _IMG_ONLY)
        # (with tight_backward the first backward records the tight graph from that iteration's reports; the full-size one is
        #  recorded only if it is ever needed -- one recording less per change of P)
        self._cap = cap
        return cap

    def _capture_backward(self, cap, pattern, used=None):
        """Backward graph for the set of outputs that receive a gradient (``pattern``: 15 booleans).  ``used``: None, or the
        batch slots in use per plain job and per composite job, baked into the launches (the TIGHT graph, see _backward)."""
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        g_in = [torch.zeros_like(o) if on else None for o, on in zip(cap.out_list, pattern)]
        outs = [o for o, on in zip(cap.out_list, pattern) if on]
        gos = [g for g in g_in if g is not None]
        inputs = cap.in_list + cap.probes
        saved = (rz.config.mode, rz.config.fixed_capacity)
        try:
            rz.config.mode, rz.config.fixed_capacity = 'capacity', list(cap.caps)
            base = self._ptr_table.data_ptr()
            rz._capture_grad_ind = {g_in[3 * i].data_ptr(): base + 8 * i for i in range(5) if g_in[3 * i] is not None}
            rz._capture_used = used
            with torch.enable_grad():
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=cap.pool):
                    grads = torch.autograd.grad(outs, inputs, grad_outputs=gos, allow_unused=True, retain_graph=True)
                # The gradients stay where the recording left them (static tensors of the graph's pool); _backward gathers
                # them into a fresh private buffer with ONE concatenation kernel after the replay -- packing inside the graph
                # AND cloning the packed buffer afterwards moved the 14 MB twice.
                unused = [gr is None for gr in grads]
                flat = [None if gr is None else gr.reshape(-1) for gr in grads]
        finally:
            rz.config.mode, rz.config.fixed_capacity = saved
            rz._capture_grad_ind = rz._capture_used = None
        self.backward_captures += 1
        if used is None:
            cap.bwd[pattern] = (g, g_in, flat, unused)
            return cap.bwd[pattern]
        cap.bwd[(pattern, 'tight')] = (g, g_in, flat, unused, used)
        return cap.bwd[(pattern, 'tight')]

    @staticmethod
    def _pack(grads, cap, flat):
        """Record the packing of ``grads`` into ``flat`` with ONE multi-tensor kernel (twenty elementwise launches cost ~40 us
        of in-graph dispatch gaps; memcpy / memset nodes are not replay-safe on this runtime, a kernel is).  Returns the list
        of "this input got no gradient" flags."""
        views, present, off = [], [], 0
        for gr, n in zip(grads, cap.sizes):
            if gr is not None:
                views.append(flat[off:off + n].view(gr.shape))
                present.append(gr)
            off += n
        if len(present) == len(grads):
            torch.cat([gr.reshape(-1) for gr in present], out=flat)
        elif present:
            torch._foreach_copy_(views, present)
        return [gr is None for gr in grads]

    # ---- per call --------------------------------------------------------------------------------------------------
    def _fill_inputs(self, cap, assets):
        with torch.no_grad():
            src = [a[k] for a in assets for k in _ASSET_KEYS + (_colour_key(a),)]
            for d, s_ in zip(cap.in_list, src):
                if tuple(d.shape) != tuple(s_.shape):
                    raise ValueError('GraphedIteration: asset tensor of shape %s, captured for %s' % (tuple(s_.shape), tuple(d.shape)))
            if self._flat_in_tight and all(s_.dtype == torch.float32 for s_ in src):
                torch.cat([s_.detach().reshape(-1) for s_ in src], out=self._flat_in)      # one kernel (14 MB: ~10 us)
            else:
                torch._foreach_copy_(cap.in_raw, [s_.detach() for s_ in src])

    def _camera(self, cam_param):
        """Write the camera block; returns (tan, check) with check = None or (pool, slot, tag) to poll after the replay."""
        cap_cam = self._cam
        f = cam_param['focal']
        pool = rz._pool()
        trusted = self._intr is not None and f is self._focal_src and getattr(f, '_version', None) == self._focal_ver
        flag = chk = None
        if self._intr is not None and not trusted and pool is not None:
            fslot, ftag, faddr = pool.take()
            flag, chk = (faddr, ftag), (pool, fslot, ftag)
        intr, checking = camera_block_device(cam_param, self.shape, cap_cam, self._intr if flag else None, flag)
        if not checking:
            self._intr, self._focal_src, self._focal_ver = intr, f, getattr(f, '_version', None)
            chk = None
        return (intr[0], intr[1]), chk

    def _focal_ok(self, chk, f):
        pool, fslot, ftag = chk
        w, b = pool.words, 4 * fslot
        t_end = time.perf_counter() + 5e-3
        while w[b + 3] != ftag and time.perf_counter() < t_end:
            pass
        if w[b + 3] != ftag:
            torch.cuda.current_stream(self.device).synchronize()
        if w[b + 3] != ftag or w[b] != 1:
            self._intr = None
            return False
        self._focal_src, self._focal_ver = f, getattr(f, '_version', None)
        return True

    def _reset_reports(self, cap):
        if rz._hdr_pool is not None:
            w = rz._hdr_pool.words
            for s in cap.slots + cap.slots_c:
                if s is not None:
                    w[4 * s[0] + 3] = 0

    def _overflowed(self, cap):
        """Wait for the header reports of the last forward replay; returns None or the capacities the renders need."""
        if not self.check or self._reports_checked:
            return None
        self._reports_checked = True
        w = rz._hdr_pool.words if rz._hdr_pool is not None else None
        needs, over = [], False
        for k, s in enumerate(cap.slots):
            if s is None or w is None:
                torch.cuda.current_stream(self.device).synchronize()
                need, ovf = self._device_header(cap, k)
            else:
                b = 4 * s[0]
                t_end = time.perf_counter() + 5e-3
                while w[b + 3] != s[1] and time.perf_counter() < t_end:
                    pass
                if w[b + 3] != s[1]:
                    torch.cuda.current_stream(self.device).synchronize()
                if w[b + 3] == s[1]:
                    need, ovf = int(w[b]), int(w[b + 1])
                else:
                    need, ovf = self._device_header(cap, k)
            needs.append(need)
            over = over or bool(ovf)
        n3 = needs if self.merge else [needs[0], needs[1], needs[3]]
        self._last_needs = (cap.sizes[0] // 3, cap.sizes[5] // 3, n3)
        return needs if over else None

    def _slot_needs(self, cap, patience=2.5e-4):
        """``num_rendered`` of every plain and composite render of the last forward replay, from their reports; None when one has
        not landed or reports are unavailable.  Waits at most ``patience`` seconds: the composites report ~0.15 ms into the forward
        graph, and a caller that reaches its backward sooner (no loss kernels in between) loses nothing by spinning that long --
        the backward graph could not start before the forward graph is through anyway."""
        if rz._hdr_pool is None:
            return None
        w = rz._hdr_pool.words
        slots = cap.slots + cap.slots_c
        if any(s is None for s in slots):
            return None
        t_end = None
        for s in slots:
            i = 4 * s[0] + 3
            while w[i] != s[1]:
                if t_end is None:
                    t_end = time.perf_counter() + patience
                elif time.perf_counter() > t_end:
                    return None
        if any(w[4 * s[0] + 1] != 0 for s in slots):
            return None
        return [int(w[4 * s[0]]) for s in slots]

    def _device_header(self, cap, k):
        raise RuntimeError('exavatar_release_amd: GraphedIteration needs pinned host memory mapped for the device '
                           '(header reports); use the eager render_iteration on this system')

    def _replay_forward(self, assets, cam_param, bg, dens, refill, loss_args=()):
        """(Re-)capture if needed, fill the static inputs, replay the forward graph.  Returns the capture."""
        dev = self.device
        # (the SH degree is baked into the captured kernel arguments: ExAvatar raises it every 1000 iterations with constant
        #  tensor shapes -- reference avatar/common/nets/module.py set_sh_degree -- and that must re-capture)
        key = tuple((tuple(a['mean_3d'].shape), _colour_key(a), tuple(a[_colour_key(a)].shape), _sh_degree(a)) for a in assets) + \
            tuple((tuple(t.shape), t.dtype) for t in loss_args)
        dens_ptrs = None if dens is None else tuple(None if t is None else t.data_ptr() for t in dens)
        for _ in range(4):
            cap = self._cap
            # camera first: tan(fov) is part of the capture key
            tan, chk = self._camera(cam_param)
            if self._bg_src is not bg or self._bg_ver != getattr(bg, '_version', None):
                with torch.no_grad():
                    torch.mul(torch.as_tensor(bg, dtype=torch.float32, device=dev).reshape(-1), 1.0, out=self._cam[35:38])
                self._bg_src, self._bg_ver = bg, getattr(bg, '_version', None)
            if cap is None or cap.key != key or cap.tan != tan or cap.dens_ptrs != dens_ptrs:
                try:
                    cap = self._capture(assets, tan, dens, key, loss_args)
                finally:
                    self._caps_hint = None
            elif refill:
                self._fill_inputs(cap, assets)
                if loss_args:
                    pairs = [(d, s_) for d, s_ in zip(cap.loss_args, loss_args) if d is not s_]
                    if pairs:
                        with torch.no_grad():
                            torch._foreach_copy_([d for d, _ in pairs], [s_.detach() for _, s_ in pairs])
            self._reset_reports(cap)
            cap.fwd.replay()
            self._serial += 1
            self._reports_checked = False
            if chk is not None and not self._focal_ok(chk, cam_param['focal']):
                # the focal length changed: derive the intrinsics again, maybe re-capture.  The replay that just ran used the
                # stale intrinsics; with the loss in the graph its backward has already updated the densification
                # statistics -- put them back (as _grow does for an overflowed replay) before the iteration runs again
                if cap.dens_backup:
                    with torch.no_grad():
                        torch._foreach_copy_(cap.dens, cap.dens_backup)
                continue
            return cap
        raise RuntimeError('exavatar_release_amd: GraphedIteration could not settle its camera intrinsics')

    @property
    def loss_inputs(self):
        """The static tensors the recorded loss reads (one per ``loss_args`` entry, None before the first call): a producer
        may write into them in place and pass them back as ``loss_args`` -- no per-call copy then."""
        return None if self._cap is None else list(self._cap.loss_args)

    def __call__(self, scene_asset, human_asset, human_asset_refined, cam_param, bg=None, scene_densify_stats=None,
                 loss_args=()):
        dev = self.device
        loss_args = tuple(loss_args)
        if loss_args and self.loss_fn is None:
            raise ValueError('GraphedIteration: loss_args without a loss_fn')
        if not all(torch.is_tensor(t) and t.device == dev for t in loss_args):
            raise ValueError('GraphedIteration: loss_args must be tensors on %s (close over everything else in loss_fn)' % dev)
        assets = (scene_asset, human_asset, human_asset_refined)
        modes = {_colour_key(a) for a in assets}
        if len(modes) != 1:
            raise ValueError('GraphedIteration: scene, human and refined human must carry the same colour input (rgb or sh)')
        if bg is None:
            bg = torch.ones(3, dtype=torch.float32, device=dev)
        dens = scene_densify_stats
        if dens is not None:
            dens = rz._check_densify(dens, int(scene_asset['mean_3d'].shape[0]), dev)
        with rz._on_device(dev):
            self._args = (assets, cam_param, bg, dens, loss_args)
            cap = self._replay_forward(assets, cam_param, bg, dens, refill=True, loss_args=loss_args)
            needs = self._overflowed(cap)
            while needs is not None:          # repaired before anybody can read the outputs
                cap = self._grow(needs)
                needs = self._overflowed(cap)
            grad = torch.is_grad_enabled() and any(a[k].requires_grad for a in assets for k in _ASSET_KEYS + (_colour_key(a),))
            loss = None
            if self.loss_fn is not None:
                # forward, loss and backward were one replay: the images are results only, the loss carries the gradients
                outs = [o.detach() for o in cap.out_list]
                if grad:
                    probes = [p.detach().requires_grad_(True) for p in cap.probes]
                    src = [a[k] for a in assets for k in _ASSET_KEYS + (_colour_key(a),)]
                    loss = _LossFn.apply(self, cap, *src, *probes)
                else:
                    probes, loss = [None] * 5, cap.loss.detach()
            elif not grad:
                outs = [o.detach() for o in cap.out_list]
                probes = [None] * 5
            else:
                # fresh leaves per iteration that alias the (never written, never read) static probes: their .grad is what
                # the reference reads after backward (avatar/main/train.py:51)
                probes = [p.detach().requires_grad_(True) for p in cap.probes]
                src = [a[k] for a in assets for k in _ASSET_KEYS + (_colour_key(a),)]
                outs = _IterFn.apply(self, cap, *src, *probes)
        self._args = None
        out = {}
        for i, name in enumerate(ITERATION_RENDERS):
            out[name] = {'img': outs[3 * i], 'depthmap': outs[3 * i + 1], 'mask': outs[3 * i + 2], 'mean_2d': probes[i],
                         'is_vis': cap.radii[name][1], 'radius': cap.radii[name][0]}
        if loss is not None:
            out['loss'] = loss
        return out

    def _grow(self, needs):
        """An instance buffer was too small: re-capture with room for ``needs`` and render the iteration again."""
        old = self._cap
        n3 = needs if self.merge else [needs[0], needs[1], needs[3]]
        c3 = old.caps if self.merge else [old.caps[0], old.caps[1], old.caps[3]]
        self._caps_hint = [max(int(n * self.growth), c) if n > c else c for n, c in zip(n3, c3)]
        self.overflow_retries += 1
        rz._record_overflow(('graphed_iteration',) + tuple(old.key), max(needs), max(old.caps), 'retried')
        assets, cam_param, bg, dens, loss_args = self._args
        if old.dens_backup:                   # (loss_fn: the overflowed replay's backward already updated the statistics)
            with torch.no_grad():
                torch._foreach_copy_(old.dens, old.dens_backup)
        self._release()
        return self._replay_forward(assets, cam_param, bg, dens, refill=False, loss_args=loss_args)

    def _backward(self, grads):
        """Replay the backward graph for the incoming image gradients; returns a private flat gradient buffer."""
        dev = self.device
        with rz._on_device(dev):
            cap = self._cap
            pattern = tuple(g is not None for g in grads)
            if not any(pattern):
                return torch.zeros(sum(cap.sizes), dtype=torch.float32, device=dev)
            # The backward blends launch one wave per 64-instance batch slot of their buffers, and those are sized with
            # head-room (plain renders) or for both sources (composites: a fifth in use): ~100 k waves per iteration that only
            # find out they have nothing to do (24 us of dispatch for the two composites alone).  This iteration's reports say
            # how many slots ARE in use; a TIGHT recording of the backward with those counts (+ 25 %) baked into its launches
            # is replayed whenever the current counts fit it, the full-size recording otherwise (reports not landed yet, or a
            # count grew past the tight one, which is then recorded again).
            entry = None
            needs = self._slot_needs(cap) if self.tight_backward else None
            if needs is not None:
                t = cap.bwd.get((pattern, 'tight'))
                if t is not None and all(n <= 64 * u for n, u in zip(needs, t[4][0] + t[4][1])):
                    entry = t[:4]
                else:
                    prev = (t[4][0] + t[4][1]) if t is not None else [0] * len(needs)
                    used = [max(int(n * 1.25) // 64 + 2, u) for n, u in zip(needs, prev)]
                    n_plain = len(cap.slots)
                    entry = self._capture_backward(cap, pattern, (used[:n_plain], used[n_plain:]))[:4]
            if entry is None:
                entry = cap.bwd.get(pattern) or self._capture_backward(cap, pattern)
            g, g_in, flat, _unused = entry
            with torch.no_grad():
                # colour gradients: the graph reads them THROUGH the pointer table, so a float32 contiguous tensor from
                # autograd is used where it lies (it outlives the replay in stream order: the caching allocator hands its
                # block only to work queued later on this stream); anything else is copied into the static buffer first.
                # Depth / mask gradients (rare) always take the copy.
                dst, src = [], []
                ptrs = (ctypes.c_void_p * 5)()
                for i, (s_, d) in enumerate(zip(grads, g_in)):
                    if d is None:
                        continue
                    direct = i % 3 == 0 and s_.dtype == torch.float32 and s_.shape == d.shape and s_.is_contiguous()
                    if i % 3 == 0:
                        ptrs[i // 3] = s_.data_ptr() if direct else d.data_ptr()
                    if not direct:
                        dst.append(d)
                        src.append(s_ if (s_.dtype == torch.float32 and s_.shape == d.shape) else s_.to(torch.float32).expand(d.shape))
                if dst:
                    torch._foreach_copy_(dst, src)
                _lib.check(_lib.load().exa_raster_store_pointers(
                    ctypes.c_void_p(self._ptr_table.data_ptr()), ptrs, 5, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            g.replay()
            with torch.no_grad():
                if not any(v is None for v in flat):
                    return torch.cat(flat)
                out = torch.zeros(sum(cap.sizes), dtype=torch.float32, device=dev)
                views = [torch.as_strided(out, (n,), (1,), o) for v, n, o in zip(flat, cap.sizes, cap.offsets) if v is not None]
                torch._foreach_copy_(views, [v for v in flat if v is not None])
                return out
