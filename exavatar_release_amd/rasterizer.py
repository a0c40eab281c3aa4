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
  ``config.fixed_capacity``; no host sync and hipGraph-capturable.
* ``'auto'`` (default): ``'capacity'`` for renders that will be differentiated and under stream
  capture, ``'exact'`` for ``torch.no_grad()`` renders.

Overflow (a capacity-mode render needed more instances than its buffer held; upstream cannot overflow
because it always takes the ``'exact'`` round trip).  The kernels latch it on the device -- the render
draws the background only, its backward would write zero gradients -- and report it through a ZERO-COPY
header: the scatter stage stores ``{needed, overflow, visible, tag}`` into 16 bytes of pinned host memory
(``ExaRasterForwardJob.host_header``) ~35 us into the forward, about when the host has finished queueing
the forward's kernels.  ONE protocol: the render's own ``forward`` polls that word before it returns
(a few microseconds, no runtime call, no synchronisation) and, per ``config.on_overflow``,

* ``'retry'`` (default): re-runs the forward with the capacity the report names, into the SAME output
  tensors, before anybody can have read them: a capacity-mode call returns exactly what ``'exact'``
  returns, images, losses computed from them and gradients alike, always (tests/test_gpu_soak.py trains
  300 iterations both ways to bit-identical parameters);
* ``'raise'``: raises ``RuntimeError``.

Nothing is ever left pending: ``backward`` finds a complete context.  (Rounds 3-4 also had deferred
protocols -- look at the report in ``backward`` or at a later call; they tripled the host state machine
for a few microseconds per call and were removed in round 5.)  Under stream capture nothing can be
polled: the report goes to a reserved slot the owner of the graph reads after each replay
(``GraphedRenderer`` / ``GraphedIteration``), or the caller checks ``read_header`` itself.
"""
import ctypes
import threading
import time
import warnings
import weakref
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
    prefiltered: bool          # accepted and ignored (include/exa_raster.h): the library always culls itself
    debug: bool


class _Config:
    mode = 'auto'             # 'auto' | 'exact' | 'capacity'
    capacity_growth = 1.5     # capacity mode: head-room over the largest D seen so far
    min_capacity = 1 << 16
    fixed_capacity = None     # capacity mode: use exactly this many instances (e.g. calibrated by a warm-up); a list /
    #                           tuple names one capacity per job of a batched call
    keep_debug = False        # developer probes: keep the workspaces of the most recent forward reachable
    on_overflow = 'retry'     # 'retry' | 'raise' (module docstring)
    overlap_composites = True     # INSIDE a stream capture: the composites' list merges run on a second stream while their sources
    #                               blend (the calls are split at EXA_RASTER_STAGE_NO_BLEND; fork / join become graph edges).
    #                               Eager calls never do this: the stream switches cost the host more than the overlap gives
    fold_composite_grads = True   # a composite's gradients for source B are handed to B's own render, whose backward adds them
    #                               inside its per-Gaussian kernel (ExaRasterBackwardJob.accumulate) instead of autograd
    #                               summing the two with one kernel per tensor (developer A/B knob; same values bit for bit)
    compose_reuse_source = True   # composite renders copy source A's pixels where source B has no entry (developer A/B knob)
    upstream_scale_grad = False   # True: dL/dscale as upstream returns it (w.r.t. scale_modifier * scale, i.e. divided
    #                               by scale_modifier); identical for the reference, which passes 1.0 (module.py:615)
    poison = False                # debug: fill every workspace with 0xFF before the kernels see it (the library promises to write
    #                               every section before it reads it; tests run under it with EXA_TEST_POISON=1)
    compiled_node = 'auto'        # single renders through the compiled autograd node (csrc/torch_binding.cpp -> _exa_torch.so: the
    #                               same C-ABI calls, arena layouts and overflow protocol as _Rasterize below at a quarter of the
    #                               host time): 'auto' = when it is built and the call is one it covers, 'off' = always the Python
    #                               node, 'require' = raise if the extension is missing


config = _Config()

_debug_last = {}  # only filled when config.keep_debug (tools/): workspaces of the most recent forward
_seen_D = {}      # (device index, P, H, W) -> largest instance capacity a call of that shape needed
overflow_events = []   # (key, needed, capacity, 'retried' | 'raised') of every overflow seen (bounded; for tests / logs)
_capture_report = None   # [(slot, tag) | None per job]: reserved header-report slots baked into the batched call being
#                          CAPTURED (set by GraphedRenderer / GraphedIteration around their capture)
_overlap = {}             # device index -> (side stream, event recorded when the sources' sorted lists are complete): set by a
#                           captured keep_keys batch, consumed by the composite call that follows it (config.overlap_composites)
_capture_report_c = None  # the same for the composite jobs of the call being captured (GraphedIteration)
_capture_used = None      # (used batch slots per plain job, per composite job) baked into the backward being RECORDED
#                           (ExaRasterBackwardJob.used_slots; GraphedIteration checks them against every replay's reports)
_capture_grad_ind = None  # {data_ptr of a static dL/dcolor buffer: device address of its pointer-table entry}: set by
#                           GraphedIteration while it RECORDS a backward graph (ExaRasterBackwardJob.dL_dcolor_indirect)
_last_handles = None     # host job records of the most recent keep_keys call (handed to rasterize_gaussians_batch's caller)
_tls = threading.local()  # .is_vis: the `radii > 0` tensors the per-Gaussian kernel of the most recent call of this thread wrote
#                           (ExaRasterForwardJob.is_vis), picked up by the renderer's output dict via take_is_vis()


def take_is_vis():
    """The boolean ``radii > 0`` tensors (one per job) of the most recent rasterizer call of this thread, written by the
    forward kernel itself -- what ``GaussianRenderer`` returns as ``is_vis`` (reference ``avatar/common/nets/layer.py``)
    without a comparison kernel per render.  One-shot: returns None when there is none or it was taken already."""
    v = getattr(_tls, 'is_vis', None)
    _tls.is_vis = None
    return v


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _workspace(nbytes, device):
    """Uninitialised byte workspace (``config.poison``: filled with 0xFF, so that a kernel reading a section nobody
    wrote sees the worst garbage instead of whatever the allocator left there)."""
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    if config.poison:
        ws.fill_(255)
    return ws


def _addr(t):
    return t.data_ptr() if t is not None else None


def _f32c(t, name, device, memo=None):
    """float32, contiguous, on ``device``.  ``memo`` (id -> converted tensor) makes K jobs that pass the SAME tensor
    object share one converted copy -- so that a batch of K views of one model still presents identical pointers to
    ``exa_raster_backward_batch(sum_shared)`` when the caller's tensors needed a ``.contiguous()`` / ``.float()``."""
    if t is None:
        return None
    if memo is not None:
        hit = memo.get(id(t))
        if hit is not None and hit[0] is t:
            return hit[1]
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a tensor' % name)
    if t.device != device:
        raise ValueError('%s is on %s, expected %s' % (name, t.device, device))
    c = t
    if c.dtype != torch.float32:
        c = c.float()
    if not c.is_contiguous():
        c = c.contiguous()
    if memo is not None:
        memo[id(t)] = (t, c)
    return c


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


_settings_cache = []      # most recent first: (key, struct, keep); the keep list pins the tensors the key's ids name


def _make_settings(rs, device, keep):
    """ctypes settings struct; tensors it points to are appended to ``keep`` so they stay alive.  Memoised on the
    identity of the four tensors + the scalars: the five renders of an iteration (and every render of a fixed camera)
    reuse one struct instead of refilling twelve fields."""
    key = (id(rs.bg), id(rs.viewmatrix), id(rs.projmatrix), id(rs.campos), rs.image_height, rs.image_width, rs.tanfovx,
           rs.tanfovy, rs.scale_modifier, rs.sh_degree, rs.prefiltered, rs.debug, device.index)
    for i, (k, s, kp) in enumerate(_settings_cache):
        if k == key and s.bg == kp[0].data_ptr() and s.viewmatrix == kp[1].data_ptr():
            if i:
                _settings_cache.insert(0, _settings_cache.pop(i))
            keep.extend(kp)
            return s
    s = _lib.ExaRasterSettings()
    s.image_height = int(rs.image_height)
    s.image_width = int(rs.image_width)
    s.tanfovx = float(rs.tanfovx)
    s.tanfovy = float(rs.tanfovy)
    s.scale_modifier = float(rs.scale_modifier)
    s.sh_degree = int(rs.sh_degree)
    s.prefiltered = int(bool(rs.prefiltered))
    s.debug = int(bool(rs.debug))
    kp = []
    cacheable = True
    for name in ('bg', 'viewmatrix', 'projmatrix', 'campos'):
        t = getattr(rs, name)
        if not (isinstance(t, torch.Tensor) and t.device == device and t.dtype == torch.float32 and t.is_contiguous()):
            if not isinstance(t, torch.Tensor):
                t = torch.as_tensor(t, dtype=torch.float32)
            t = t.to(device=device, dtype=torch.float32).contiguous()
            cacheable = False               # the converted copy is ours: its id says nothing about the caller's object
        kp.append(t)
        setattr(s, name, t.data_ptr())
    keep.extend(kp)
    if cacheable:
        _settings_cache.insert(0, (key, s, kp))
        del _settings_cache[8:]
    return s


# ---- zero-copy header reports ------------------------------------------------------------------------------------
class _HdrPool:
    """16-byte slots in pinned host memory the scatter kernel writes its header report into: a ring of N slots for eager
    calls (one per render job, composite or focal-length flag; a report is consumed inside the call that took its slot,
    so a slot comes round again long after its report was read) and RESERVED slots outside the ring for reports that are
    baked into a captured hipGraph (``GraphedRenderer``, ``GraphedIteration``): a graph rewrites its slot on every
    replay for as long as it lives, so it must never be dealt to anybody else."""
    N = 2048
    RESERVED = 1024
    COMPILED = 8       # behind the reserved slots: the compiled autograd node's own small ring (a report is consumed inside its call)

    def __init__(self):
        total = self.N + self.RESERVED
        self.compiled_first = total
        total += self.COMPILED
        self.buf = torch.zeros((total, 4), dtype=torch.int32, pin_memory=True)
        dp = ctypes.c_void_p()
        _lib.check(_lib.load().exa_raster_host_device_pointer(ctypes.c_void_p(self.buf.data_ptr()), ctypes.byref(dp)))
        self.dev_base = int(dp.value)
        self.words = (ctypes.c_uint32 * (4 * total)).from_address(self.buf.data_ptr())
        self.next = 0
        self.tag = 1
        self.free_reserved = list(range(self.compiled_first - 1, self.N - 1, -1))

    def _next_tag(self):
        tag = self.tag
        self.tag = tag + 1 if tag < 0x7ffffff0 else 1
        return tag

    def take(self):
        """(slot, tag, device address) of the next ring slot."""
        i = self.next
        self.next = (i + 1) % self.N
        return i, self._next_tag(), self.dev_base + 16 * i

    def reserve(self):
        """(slot, tag, device address) of a slot outside the ring, held until :meth:`release`."""
        if not self.free_reserved:
            raise RuntimeError('exavatar_release_amd: all %d reserved header-report slots are taken: too many live '
                               'GraphedRenderer / GraphedIteration objects (close() the ones no longer used)' % self.RESERVED)
        i = self.free_reserved.pop()
        self.words[4 * i + 3] = 0
        return i, self._next_tag(), self.dev_base + 16 * i

    def release(self, slot):
        if slot >= self.N and slot not in self.free_reserved:
            self.free_reserved.append(slot)


_hdr_pool = None
_hdr_pool_failed = False


def _pool():
    """The pool, or None when pinned host memory cannot be mapped for the device (then the header is read back with a
    16-byte copy + a stream synchronisation per call)."""
    global _hdr_pool, _hdr_pool_failed
    if _hdr_pool is None and not _hdr_pool_failed:
        try:
            _hdr_pool = _HdrPool()
        except Exception:  # noqa: BLE001
            _hdr_pool_failed = True
    return _hdr_pool


def _await_report(slot, tag, stream):
    """(needed capacity, overflow flag) of the report with ``tag`` in pool slot ``slot``.  The scatter stage writes it
    ~35 us into the forward: the host spins on the word (no runtime call); a stream that is far behind is waited for."""
    w, i = _hdr_pool.words, 4 * slot + 3
    if w[i] != tag:
        t_end = time.perf_counter() + 2e-3
        while w[i] != tag and time.perf_counter() < t_end:
            pass
        if w[i] != tag:
            stream.synchronize()
            if w[i] != tag:
                raise RuntimeError('exavatar_release_amd: the header report of a render never arrived')
    return int(w[i - 3]), int(w[i - 2])


def _landed_need(report):
    """``num_rendered`` of a ``(slot, tag)`` report that has landed and did not overflow, else None; never waits."""
    if report is None or _hdr_pool is None:
        return None
    w, b = _hdr_pool.words, 4 * report[0]
    if w[b + 3] != report[1] or w[b + 1] != 0:
        return None
    return int(w[b])


def _record_overflow(key, need, cap, how):
    overflow_events.append((key, need, cap, how))
    del overflow_events[:-64]


def _note(key, need):
    if need > _seen_D.get(key, 0):
        _seen_D[key] = need


def _overflow_error(need, cap):
    return RuntimeError('exavatar_release_amd: tile-instance buffer overflow (needed %d, capacity %d). '
                        'Use config.on_overflow="retry", config.mode="exact" or raise config.capacity_growth.' % (need, cap))


def _rerender(j, need, store_ctx, device):
    """Run job ``j``'s forward again with room for ``need`` instances, into the same output tensors."""
    lib = _lib.load()
    cap = (max(int(need), 64) + 63) // 64 * 64
    j.capacity = cap
    j.ws = _workspace(j.gb + j.tb + int(_sizes(j.P, j.W, j.H, cap).bin_bytes), device)
    j.bins = None
    j.geom_ptr = j.ws.data_ptr()
    j.tile_ptr = j.geom_ptr + j.gb
    j.bin_ptr = j.tile_ptr + j.tb
    arr = (_lib.ExaRasterForwardJob * 1)()
    _fill_forward_job(arr[0], j)
    _lib.check(lib.exa_raster_forward_batch(arr, 1, int(store_ctx), _stream_ptr(device)))


def _settle(jobs, reports, store_ctx, device, stream):
    """Read the header reports of a capacity-mode call that was just queued (``reports``: one ``(slot, tag)`` per job, or
    None without a pool: the headers are read back) and deal with overflowed jobs (``config.on_overflow``) before the
    call's outputs leave ``forward``.  Every report is consumed before anything is raised."""
    if reports is None:
        rows = [j.ws[j.gb:j.gb + 16].view(torch.int32) for j in jobs]
        hdr = (rows[0] if len(jobs) == 1 else torch.stack(rows)).cpu().view(len(jobs), 4)
        got = [(int(hdr[k, 0]), int(hdr[k, 1])) for k in range(len(jobs))]
    else:
        got = [_await_report(r[0], r[1], stream) for r in reports]
    err = None
    for j, (need, overflow) in zip(jobs, got):
        _note(j.key, need)
        j.need = need                # (the backward launches one wave per batch slot IN USE: ExaRasterBackwardJob.used_slots)
        if not overflow:
            continue
        if config.on_overflow == 'raise':
            _record_overflow(j.key, need, j.capacity, 'raised')
            err = err or _overflow_error(need, j.capacity)
            continue
        old = j.capacity
        _rerender(j, need, store_ctx, device)
        _record_overflow(j.key, need, old, 'retried')
    if err is not None:
        raise err


HEADER_FIELDS = ('num_rendered', 'overflow', 'max_tile_list', 'num_visible', 'num_instances', 'active_cells',
                 'num_tile_instances')


def read_header(tile_ws):
    """(num_rendered, overflow, entries, num_visible, num_instances, active_cells, num_tile_instances) of a tile
    workspace tensor (synchronises).  ``num_tile_instances`` = upstream's num_rendered (16x16 tile instances)."""
    return tuple(int(v) for v in tile_ws[:28].view(torch.int32).cpu())


def last_header():
    """Header of the most recent forward; needs ``config.keep_debug = True`` (developer probes only)."""
    return read_header(_debug_last['tile'])


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
                 'gb', 'tb', 'keep_keys', 'device', 'is_vis', 'stash', 'token_ref', 'need')


_F32 = torch.float32
_PLANES = [3, 1, 1]         # colour | depth | alpha planes of one output arena


def _fill_forward_job(a, j, report=None):
    a.settings = ctypes.pointer(j.settings)
    a.P, a.sh_M = j.P, j.sh_M
    a.means3D, a.shs, a.colors_precomp = _addr(j.means3D), _addr(j.sh), _addr(j.colors)
    a.opacities, a.scales, a.rotations = _addr(j.opac), _addr(j.scales), _addr(j.rot)
    a.cov3D_precomp = _addr(j.cov)
    a.radii = j.radii.data_ptr()
    a.is_vis = j.is_vis.data_ptr()
    a.geom_ws, a.tile_ws, a.bin_ws, a.capacity = j.geom_ptr, j.tile_ptr, j.bin_ptr, j.capacity
    base = j.planes.data_ptr()
    a.out_color, a.out_depth, a.out_alpha = base, base + 12 * j.H * j.W, base + 16 * j.H * j.W
    a.keep_sorted_keys = 1 if j.keep_keys else 0
    if report is not None:           # (slot, tag) of a pool slot
        a.host_header, a.header_tag = _hdr_pool.dev_base + 16 * report[0], report[1]
    else:
        a.host_header, a.header_tag = None, 0


def _grad_in(g, shape, device):
    """Incoming image gradient as a contiguous float32 tensor of ``shape`` (the common case returns ``g`` itself)."""
    if g is None:
        return None
    if g.dtype is _F32 and g.shape == shape and g.is_contiguous():
        return g
    g = g.to(dtype=_F32, device=device).expand(shape)
    return g if g.is_contiguous() else g.contiguous()


N_IN = 8      # tensor arguments per job: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D


def _grad_pattern(*wanted):
    """Which of (means3D, sh, colors, opacity, scales, rotations, cov3D) get a gradient, as a tuple of bools."""
    return tuple(bool(w) for w in wanted)


class _Rasterize(torch.autograd.Function):
    """K renders, one launch per pipeline stage.  apply(K, settings, grad_enabled, shared, densify, frozen, opts, *tensors[8 K]).
    ``opts``: None or a dict -- ``keep_keys``: the renders will be sources of composite renders (:class:`_Compose`): their
    sorts keep the sorted 64-bit keys, and the host records of the jobs are published in ``_last_handles``.
    ``densify``: None or a K-list of None / (xyz_grad_accum, track_cnt, radius_max) tensors updated IN PLACE by that
    render's backward (fused densification statistics, include/exa_raster.h).
    ``frozen``: None or a K-list of None / 8-tuples in the order of ``_IN_NAMES``: a CONSTANT prefix of Gaussians that
    render k blends in front of / behind its own (the detached scene of ExAvatar's composite renders,
    ``avatar/main/model.py:119-126``).  The render sees ``cat(prefix, own)``; the backward returns gradients for the own
    rows only and skips all work on the prefix (``ExaRasterBackwardJob.grad_first``)."""

    @staticmethod
    def forward(ctx, K, settings, grad_enabled, shared, densify, frozen, opts, *tensors):
        global _last_handles
        lib = _lib.load()
        keep_keys = bool(opts and opts.get('keep_keys'))
        device = tensors[0].device
        if device.type != 'cuda':
            raise RuntimeError('exavatar_release_amd: the rasterizer runs on a ROCm device only '
                               '(got %s); there is no CPU path' % device)
        # autograd is off inside forward(): the caller samples torch.is_grad_enabled() (a render under no_grad must
        # not pay for the backward context although GaussianRenderer's mean_2d probe requires grad)
        need_ctx = bool(grad_enabled) and any(ctx.needs_input_grad)
        memo = {} if K > 1 else None
        jobs = []
        for k in range(K):
            m3, _m2, sh, col, op, sc, rot, cov = tensors[N_IN * k: N_IN * (k + 1)]
            j = _Job()
            j.rs = settings[k]
            j.H, j.W = int(j.rs.image_height), int(j.rs.image_width)
            fz = frozen[k] if frozen is not None else None
            j.nF = int(fz[0].shape[0]) if fz is not None else 0

            def inp(t, i, name):
                t = _f32c(t, name, device, memo)
                if fz is None:
                    return t
                f = _f32c(fz[i], name + ' (constant prefix)', device, memo)
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
            j.keep_keys, j.device = keep_keys, device
            jobs.append(j)
        if shared and K > 1:
            # "K views of the same Gaussians" is decided on what the kernels will see: the converted tensors
            j0 = jobs[0]
            shared = all(jk.P == j0.P and all(_addr(getattr(jk, n)) == _addr(getattr(j0, n))
                                              for n in ('means3D', 'sh', 'colors', 'opac', 'scales', 'rot', 'cov'))
                         for jk in jobs[1:])

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
            arr = (_lib.ExaRasterForwardJob * K)()
            # capacity mode outside a capture: every job reports its header into a pinned-host slot (module docstring)
            polled = mode == 'capacity' and not capturing
            reports = [_hdr_pool.take()[:2] for _ in range(K)] if polled and _pool() is not None else None
            for k, j in enumerate(jobs):
                j.keep = []
                j.settings = _make_settings(j.rs, device, j.keep)
                j.planes = torch.empty((5, j.H, j.W), dtype=_F32, device=device)   # colour | depth | alpha
                j.radii = torch.empty((j.P,), dtype=torch.int32, device=device)
                j.is_vis = torch.empty((j.P,), dtype=torch.bool, device=device)      # (one byte each: 0 / 1 from the kernel)
                j.stash = j.token_ref = j.need = None
                sz = _sizes(j.P, j.W, j.H, 0)
                j.gb, j.tb = int(sz.geom_bytes), int(sz.tile_bytes)
                j.bins = None
                if mode == 'capacity':
                    if config.fixed_capacity is not None:
                        fc = config.fixed_capacity
                        cap = int(fc[k] if isinstance(fc, (list, tuple)) else fc)
                    else:
                        cap = max(int(_seen_D[j.key] * config.capacity_growth), config.min_capacity)
                    j.capacity = (cap + 63) // 64 * 64
                    # ONE arena per render: splat records | tile workspace | bin workspace
                    j.ws = _workspace(j.gb + j.tb + int(_sizes(j.P, j.W, j.H, j.capacity).bin_bytes), device)
                    j.bin_ptr = j.ws.data_ptr() + j.gb + j.tb
                else:
                    j.capacity = 0
                    j.ws = _workspace(j.gb + j.tb, device)
                    j.bin_ptr = None
                j.geom_ptr = j.ws.data_ptr()
                j.tile_ptr = j.geom_ptr + j.gb
                rep = reports[k] if reports is not None else None
                if capturing and _capture_report is not None and k < len(_capture_report):
                    rep = _capture_report[k]          # a reserved slot, read by the owner of the graph after each replay
                _fill_forward_job(arr[k], j, rep)

            if mode == 'exact':
                _lib.check(lib.exa_raster_forward_bin_batch(arr, K, stream))
                rows = [j.ws[j.gb:j.gb + 16].view(torch.int32) for j in jobs]
                hdr = (rows[0] if K == 1 else torch.stack(rows)).cpu().view(K, 4)        # D2H + sync, as upstream does
                for k, j in enumerate(jobs):
                    j.capacity = max(int(hdr[k, 0]), 64)          # header reports whole 64-instance batch slots
                    j.need = int(hdr[k, 0])
                    _note(j.key, int(hdr[k, 0]))
                    j.bins = _workspace(_sizes(j.P, j.W, j.H, j.capacity).bin_bytes, device)
                    j.bin_ptr = j.bins.data_ptr()
                    arr[k].bin_ws, arr[k].capacity = j.bin_ptr, j.capacity
                _lib.check(lib.exa_raster_forward_render_batch(arr, K, int(need_ctx), stream))
            else:
                if capturing and keep_keys and config.overlap_composites:
                    # lists first, an event, then the blend: the composites that follow merge these lists on a side stream
                    # while this blend runs (_Compose.forward)
                    _lib.check(lib.exa_raster_forward_batch(arr, K, int(need_ctx) | _lib.STAGE_NO_SORT, stream))
                    ev_binned = torch.cuda.Event()
                    ev_binned.record(stream_obj)
                    _lib.check(lib.exa_raster_forward_batch(arr, K, int(need_ctx) | _lib.STAGE_SORT_ONLY, stream))
                    ev_sorted = torch.cuda.Event()
                    ev_sorted.record(stream_obj)
                    _overlap[device.index] = (ev_binned, ev_sorted)
                    _lib.check(lib.exa_raster_forward_batch(arr, K, int(need_ctx) | _lib.STAGE_BLEND_ONLY, stream))
                else:
                    _overlap.pop(device.index, None)
                    _lib.check(lib.exa_raster_forward_batch(arr, K, int(need_ctx), stream))
                if polled:
                    # the reports land about when the last launch above was queued: an overflowed render is repaired (or
                    # raised about) HERE, before its outputs leave this function
                    _settle(jobs, reports, need_ctx, device, stream_obj)
        if config.keep_debug:
            j = jobs[-1]
            _debug_last['tile'] = j.ws[j.gb:j.gb + j.tb]
            _debug_last['geom'] = j.ws[:j.gb]
            _debug_last['bin'] = j.bins if getattr(j, 'bins', None) is not None else j.ws[j.gb + j.tb:]
            _debug_last['capacity'] = j.capacity
        if keep_keys:
            _last_handles = jobs
        _tls.is_vis = [j.is_vis for j in jobs]
        ctx.need_ctx = need_ctx
        outs = []
        for j in jobs:
            c, d, a = torch.split_with_sizes(j.planes, _PLANES)       # (Tensor.split is a Python wrapper: ~4 us)
            outs += [c, j.radii, d, a]
        if keep_keys:
            # One more (empty) differentiable output: composites of these renders take it as an input, which orders their
            # backward BEFORE this node's and guarantees that this node's backward runs whenever theirs did -- the
            # composites can then leave their gradients for source B with B's job (``_Job.stash``) and this backward adds
            # them in its per-Gaussian kernel (config.fold_composite_grads).
            outs.append(torch.empty(0, dtype=_F32, device=device))
        if need_ctx:
            ctx.K = K
            ctx.densify = densify
            ctx.shared = bool(shared) and K > 1
            ctx.jobs = jobs
            ctx.has = [tuple(t is not None for t in (j.sh, j.colors, j.scales, j.rot, j.cov)) for j in jobs]
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
        need = ctx.needs_input_grad[7:]
        # a render none of whose images received a gradient does no work and returns None, as its own node would if it had
        # been rendered by a call of its own (its backward would not run at all); K views summed in the kernel: all or none
        dead = [grads[4 * k] is None and grads[4 * k + 2] is None and grads[4 * k + 3] is None for k in range(K)]
        if ctx.shared and not all(dead):
            dead = [False] * K
        n_live = K - sum(dead)
        arr = (_lib.ExaRasterBackwardJob * max(n_live, 1))()
        keep, ret, late, pos, sides = [], [None, None, None, None, None, None, None], [], 0, set()
        with _on_device(device):
            for k in range(K):
                j = ctx.jobs[k]
                P, H, W, sh_M, nF = j.P, j.H, j.W, j.sh_M, j.nF
                Pg = P - nF                               # rows of every gradient array (constant prefix excluded)
                has_sh, has_col, has_sc, has_rot, has_cov = ctx.has[k]
                means3D, sh, col, opac, scales, rot, cov, radii = saved[8 * k: 8 * k + 8]
                g_color, g_depth, g_alpha = grads[4 * k], grads[4 * k + 2], grads[4 * k + 3]
                if dead[k]:
                    st, j.stash = j.stash, None
                    if st is not None and len(st) > 3:
                        sides.add(st[3])
                    if st is None:
                        ret += [None] * N_IN
                    else:                              # only the composites of this render were differentiated
                        h3, hsh, hcol, hop, hsc, hrot, hcov = st[2]
                        ret += [h3, None, hsh, hcol, hop, hsc, hrot, hcov]
                    continue

                g_color = _grad_in(g_color, (3, H, W), device)
                if g_color is None:
                    g_color = torch.zeros((3, H, W), dtype=_F32, device=device)
                if g_depth is not None:
                    g_depth = _grad_in(g_depth, (1, H, W), device)
                if g_alpha is not None:
                    g_alpha = _grad_in(g_alpha, (1, H, W), device)
                nd = need[N_IN * k: N_IN * (k + 1)]
                own = (not ctx.shared) or k == 0          # shared: job 0's outputs receive the sum over the K views
                # ONE arena for the small per-Gaussian gradients of this job (3 + 3 + 3 + 1 + 3 + 4 + 6 floats per row at
                # most) instead of up to seven allocator calls; dL/dsh (up to 48 floats per row) stays its own tensor.
                # AccumulateGrad adopts a contiguous view as `.grad` like any other tensor.
                want = ((own and nd[0], 3), (nd[1], 3), (own and has_col and nd[3], 3), (own and nd[4], 1),
                        (own and has_sc and nd[5], 3), (own and has_rot and nd[6], 4), (own and has_cov and nd[7], 6))
                want_sh = own and has_sh and nd[2]
                # gradients a composite render left for these Gaussians (_Compose.backward): this call adds its own to them
                st, j.stash = j.stash, None
                if st is not None and len(st) > 3:
                    sides.add(st[3])              # (recorded on a side stream inside a capture: joined before it is read)
                fold = st is not None and not ctx.shared and nF == 0 and \
                    st[1] == _grad_pattern(want[0][0], want_sh, want[2][0], want[3][0], want[4][0], want[5][0], want[6][0])
                if fold:
                    d_means3D, d_sh, d_colors, d_opac, d_scales, d_rot, d_cov = st[2]
                    d_means2D = torch.empty((Pg, 3), dtype=_F32, device=device) if nd[1] else None
                else:
                    widths = [w for on, w in want if on]
                    pieces = iter(torch.split_with_sizes(torch.empty(Pg * sum(widths), dtype=_F32, device=device), [Pg * w for w in widths])) \
                        if widths else iter(())
                    d_means3D, d_means2D, d_colors, d_opac, d_scales, d_rot, d_cov = \
                        [next(pieces).view(Pg, w) if on else None for on, w in want]
                    d_sh = torch.empty((Pg, sh_M, 3), dtype=_F32, device=device) if want_sh else None
                    if st is not None:          # (a pattern this kernel path does not add in place: summed below)
                        late += [(d, h) for d, h in zip((d_means3D, d_sh, d_colors, d_opac, d_scales, d_rot, d_cov), st[2])
                                 if d is not None and h is not None]
                grad_ws = _workspace(_sizes(P, W, H, j.capacity).grad_bytes, device)
                keep += [g_color, g_depth, g_alpha, grad_ws]
                a = arr[pos]
                pos += 1
                a.settings = ctypes.pointer(j.settings)   # built in forward; its tensors are kept alive by j.keep
                a.P, a.sh_M = P, sh_M
                a.means3D = means3D.data_ptr()
                a.shs = sh.data_ptr() if has_sh else None
                a.colors_precomp = col.data_ptr() if has_col else None
                a.opacities = opac.data_ptr()
                a.scales = scales.data_ptr() if has_sc else None
                a.rotations = rot.data_ptr() if has_rot else None
                a.cov3D_precomp = cov.data_ptr() if has_cov else None
                a.radii = radii.data_ptr()
                a.geom_ws, a.tile_ws, a.bin_ws = j.geom_ptr, j.tile_ptr, j.bin_ptr
                a.capacity = j.capacity
                a.dL_dcolor, a.dL_ddepth, a.dL_dalpha = g_color.data_ptr(), _addr(g_depth), _addr(g_alpha)
                if _capture_grad_ind is not None:
                    a.dL_dcolor_indirect = _capture_grad_ind.get(g_color.data_ptr())
                a.grad_ws = grad_ws.data_ptr()
                a.dL_dmeans2D, a.dL_dmeans3D, a.dL_dcolors = _addr(d_means2D), _addr(d_means3D), _addr(d_colors)
                a.dL_dopacity, a.dL_dscales, a.dL_drotations = _addr(d_opac), _addr(d_scales), _addr(d_rot)
                a.dL_dsh, a.dL_dcov3D = _addr(d_sh), _addr(d_cov)
                dens = ctx.densify[k] if ctx.densify is not None else None
                if dens is not None:
                    if d_means2D is None:          # the statistics need the screen-space gradient: compute it anyway
                        d_tmp = torch.empty((Pg, 3), dtype=_F32, device=device)
                        keep.append(d_tmp)
                        a.dL_dmeans2D = d_tmp.data_ptr()
                    a.densify_grad_accum, a.densify_track_cnt, a.densify_radius_max = [_addr(t) for t in dens]
                a.grad_first = nF
                a.accumulate = 1 if fold else 0
                a.used_slots = (j.need + 63) // 64 if j.need else 0
                if _capture_used is not None and k < len(_capture_used[0]):
                    a.used_slots = int(_capture_used[0][k])
                if config.upstream_scale_grad and d_scales is not None and float(j.rs.scale_modifier) != 1.0:
                    keep.append((d_scales, float(j.rs.scale_modifier)))
                ret += [d_means3D, d_means2D, d_sh, d_colors, d_opac, d_scales, d_rot, d_cov]
            if sides:
                # a composite's backward runs on a side stream (config.overlap_composites, inside a capture): this batch's
                # blend backward overlaps it, the join comes before the chain rule that adds to the composite's gradients
                main = torch.cuda.current_stream(device)
                if n_live:
                    _lib.check(lib.exa_raster_backward_batch(arr, n_live, int(ctx.shared) | _lib.STAGE_BLEND_ONLY, _stream_ptr(device)))
                for side in sides:
                    main.wait_stream(side)
                if n_live:
                    _lib.check(lib.exa_raster_backward_batch(arr, n_live, int(ctx.shared) | _lib.STAGE_NO_BLEND, _stream_ptr(device)))
            elif n_live:
                _lib.check(lib.exa_raster_backward_batch(arr, n_live, int(ctx.shared), _stream_ptr(device)))
            for item in keep:
                if isinstance(item, tuple):        # upstream's dL/dscale quirk: gradient w.r.t. (modifier * scale)
                    item[0].div_(item[1])
            for d, h in late:
                d.add_(h)
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


def _check_densify_aliasing(dens, shared):
    """Two jobs of one batch must not update the same statistics array (plain read-modify-writes by concurrently running
    workgroups) -- except K views of the SAME Gaussians that all name the same three arrays: the kernel then sums the K
    views' statistics and writes once per Gaussian (``exa_raster_backward_batch``, ``sum_shared``)."""
    live = [d for d in dens if d is not None and any(t is not None for t in d)]
    if len(live) < 2:
        return
    ptrs = [tuple(_addr(t) for t in d) for d in live]
    if shared and len(live) == len(dens) and all(p == ptrs[0] for p in ptrs):
        return
    seen = set()
    for p in ptrs:
        for a in p:
            if a is None:
                continue
            if a in seen:
                raise ValueError('densify_stats: two renders of one batch update the same tensor; give every render its '
                                 'own statistics, or -- for K views of the same Gaussians -- pass the same three tensors '
                                 'to all of them')
        seen.update(a for a in p if a is not None)


_compiled = None      # the compiled autograd node (csrc/torch_binding.cpp): module, or False when it cannot be used
compiled_calls = 0    # renders the compiled node has taken (tests / logs)


def _compiled_node():
    """``_exa_torch`` initialised against the loaded library and its header-report slots; False if unavailable."""
    global _compiled
    if _compiled is None:
        _compiled = False
        try:
            from . import _exa_torch
            pool = _pool()
            if pool is None:
                raise RuntimeError('pinned host memory cannot be mapped for the device')
            _lib.load()
            _exa_torch.init(_lib.LIB_PATH, pool.buf.data_ptr() + 16 * pool.compiled_first, pool.dev_base + 16 * pool.compiled_first,
                            pool.COMPILED)
            _compiled = _exa_torch
        except Exception as e:  # noqa: BLE001
            if config.compiled_node == 'require':
                _compiled = None
                raise
            warnings.warn('exavatar_release_amd: the compiled autograd node is not available (%s): single renders go through '
                          'the Python node, ~0.2 ms of host time slower per fwd + bwd (python -m exavatar_release_amd.build)' % e,
                          RuntimeWarning)
    return _compiled


def _rasterize_compiled(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, dens):
    """One render through the compiled node, or None when this call is not one it covers (the Python node takes it)."""
    cfg = config
    mode = cfg.mode
    if cfg.on_overflow != 'retry' or cfg.keep_debug or cfg.upstream_scale_grad or mode not in ('auto', 'exact', 'capacity') \
            or not means3D.is_cuda:
        return None
    node = _compiled if _compiled is not None else _compiled_node()
    if not node:
        return None
    key = (means3D.device.index, means3D.shape[0], rs[0], rs[1])
    cap = 0                       # 'exact': two stages with a host round trip in between, as upstream does
    if mode != 'exact':
        cap = cfg.fixed_capacity
        if cap is None:
            seen = _seen_D.get(key)       # (first call of a shape: measured exactly once)
            cap = 0 if seen is None else max(int(seen * cfg.capacity_growth), cfg.min_capacity)
        elif isinstance(cap, (list, tuple)):
            cap = cap[0]
    res = node.rasterize(rs, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, int(cap),
                         mode == 'auto', cfg.poison, dens)
    if res is None:
        return None
    global compiled_calls
    compiled_calls += 1
    color, radii, depth, alpha, is_vis, need, overflowed = res
    if need > _seen_D.get(key, 0):
        _seen_D[key] = need
    if overflowed:
        _record_overflow(key, need, overflowed, 'retried')
    _tls.is_vis = [is_vis]
    return color, radii, depth, alpha


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, densify_stats=None):
    """``densify_stats``: optional ``(xyz_grad_accum, track_cnt, radius_max)`` float32 tensors of P elements that THIS
    render's backward updates in place (fused densification statistics, see ``densify.track_densify_stats``)."""
    dens = None if densify_stats is None else [_check_densify(densify_stats, int(means3D.shape[0]), means3D.device)]
    if config.compiled_node != 'off' and isinstance(raster_settings, tuple) and _capture_report is None:
        out = _rasterize_compiled(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                  raster_settings, dens[0] if dens else None)
        if out is not None:
            return out
    return _Rasterize.apply(1, (raster_settings,), torch.is_grad_enabled(), False, dens, None, None, means3D, means2D, sh,
                            colors_precomp, opacities, scales, rotations, cov3Ds_precomp)


_IN_NAMES = ('means3D', 'means2D', 'shs', 'colors_precomp', 'opacities', 'scales', 'rotations', 'cov3D_precomp')


def rasterize_gaussians_batch(jobs, keep_keys=False):
    """K renders in one launch per pipeline stage.

    ``jobs``: sequence of dicts with the keyword arguments of ``GaussianRasterizer.forward`` plus
    ``raster_settings`` (and optionally ``densify_stats``, see :func:`rasterize_gaussians`, and ``frozen``: a dict with
    the same tensor keywords holding a CONSTANT prefix of Gaussians -- the render blends ``cat(prefix, own)``, ``radii``
    covers both, gradients / ``means2D`` / ``densify_stats`` cover the job's own Gaussians only and the backward does
    no work for the prefix).  Returns a list of ``(color, radii, depth, alpha)`` tuples, bit-identical to K single
    calls.  When every job passes the SAME tensor objects for the Gaussians (K views of one model), the backward
    sums the K views' gradients inside the per-Gaussian kernel (one thread walks the K views) instead of letting
    autograd add K gradient tensors; such a batch may also share ONE set of ``densify_stats`` tensors.
    ``keep_keys=True``: the renders will be sources of composite renders (:func:`rasterize_composites`); returns
    ``(outputs, handles)`` with one opaque handle per job.
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
        _check_densify_aliasing(dens, shared)
    global _last_handles
    _last_handles = None
    outs = _Rasterize.apply(K, tuple(j['raster_settings'] for j in jobs), torch.is_grad_enabled(), shared, dens, frozen,
                            {'keep_keys': True} if keep_keys else None, *flat)
    res = [tuple(outs[4 * k: 4 * k + 4]) for k in range(K)]
    if keep_keys:
        handles, _last_handles = _Handles(_last_handles), None
        handles.token = outs[4 * K]
        ref = weakref.ref(handles.token)       # (weak: job -> token -> grad_fn -> ctx -> job would keep the workspaces alive)
        for j in handles:
            j.token_ref = ref
        return res, handles
    return res


class _Handles(list):
    """Handles of a ``keep_keys`` batch; ``token``: see :class:`_Compose`."""
    token = None


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


from .composite import _CJob, _Compose, _compose_launch, rasterize_composites      # noqa: E402,F401  (composite renders: composite.py)
