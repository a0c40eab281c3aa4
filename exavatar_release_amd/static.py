"""``StaticRender``: one render configuration driven straight through the C ABI (``include/exa_raster.h``) with STATIC
storage -- the caller's input tensors, per render slot one set of workspaces, output images and (training) gradient arrays,
all allocated once -- and the job structures of every camera marshalled once.  A forward is ONE ``ctypes`` call
(``exa_raster_forward_batch``: five kernel launches), a backward ONE (``exa_raster_backward_batch``: two launches);
nothing is allocated, converted or filled per call.  It is the shape of a native (C++) trainer's inner loop -- what the C ABI
is for -- and what ``bench.py --launch abi`` times: the host spends ~30 us per step, the cameras are read where they lie
(every view's settings point into a resident table of views), the next step's launches queue up behind the running one.

Not autograd: gradients land in the arrays given to (or allocated by) the object; combine with ``torch`` by treating
``.grads`` as the ``.grad`` of the parameters.  Same kernels, same results as the autograd surface
(``GaussianRasterizer``, reference ``avatar/common/nets/module.py:632-640``) bit for bit (``tests/test_gpu_static.py``).

**Slots** (``slots=S``): S renders in flight, each with its own workspaces, images and HIP stream, sharing the inputs.  The
binning chain of a render is latency-bound and its blends throughput-bound, so the views of a rank's shard (25 of them per
epoch in the reference's configuration) overlap well: S = 4 renders ~45 % more views per second than one at a
time (7 200 -> 10 400-10 600 it/s on C3).  ``backward(..., accumulate=True, after=slot)`` chains the small per-Gaussian kernels of a group of slots so that
their gradients are ADDED into one set of arrays in a fixed order (the sum a trainer with a batch of views needs, bit-identical
to rendering the views one after the other) while the heavy blend backwards overlap freely.

**Life cycle**: ``rebind(...)`` after the Gaussian count changed (the reference densifies / prunes every 100 iterations,
``avatar/main/model.py:279-292``): reallocates what depends on P, keeps cameras and image-sized buffers, re-marshals the jobs.

**Overflow** of a slot's instance buffer (``capacity``), per ``on_overflow``:

* ``'repair'`` (default): every forward polls its own zero-copy header report (written by the scatter stage ~35 us into the
  render; a spin on a pinned-host word, no runtime call) before it returns; an overflowed render is re-rendered THERE into the
  same images with the capacity the report names (+ 25 %), the slot keeps the larger buffers.  The host stays at most one
  forward ahead of the device, which costs nothing while its ~30 us per step are below the device's step.
* ``'raise'``: nothing is polled; the report of a slot's previous render is read right before its next forward and at
  :meth:`check`, and an overflow raises there.  The images AND gradients of such a render are invalid (background only,
  zeros): with this policy do not consume or all-reduce gradients before ``check()`` has passed.
"""
import ctypes
import math

import torch

from . import _lib
from . import rasterizer as _rz

_F32 = torch.float32


def _addr(t):
    return None if t is None else t.data_ptr()


def required_capacity(means3D, opacities, scales, rotations, colors_precomp=None, shs=None, *, settings):
    """Instances (whole 64-entry batch slots) the renders of ``settings`` (one ``GaussianRasterizationSettings`` or a list)
    need: the binning stage alone through ``exa_raster_forward_bin``, one header read-back per camera (synchronises; set-up)."""
    lib = _lib.load()
    device = means3D.device
    if hasattr(settings, 'image_height'):          # one settings tuple (a NamedTuple IS a tuple)
        settings = [settings]
    P = int(means3D.shape[0])
    H, W = int(settings[0].image_height), int(settings[0].image_width)
    sz = _lib.workspace_sizes(P, W, H, 0)
    need = 0
    with torch.cuda.device(device):
        geom = _rz._workspace(sz.geom_bytes, device)
        tile = _rz._workspace(sz.tile_bytes, device)
        radii = torch.empty(P, dtype=torch.int32, device=device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        for rs in settings:
            keep = []
            st = _rz._make_settings(rs, device, keep)
            _lib.check(lib.exa_raster_forward_bin(ctypes.byref(st), P, int(shs.shape[1]) if shs is not None else 0,
                                                  _addr(means3D), _addr(shs), _addr(colors_precomp), _addr(opacities), _addr(scales),
                                                  _addr(rotations), None, _addr(radii), _addr(geom), _addr(tile), stream))
            need = max(need, _rz.read_header(tile)[0])
    return need


class _Slot:
    """One render in flight: workspaces, images, stream, the report slot of its forwards."""
    __slots__ = ('index', 'stream', 'stream_ptr', 'geom', 'tile', 'bin', 'grad_ws', 'planes', 'color', 'depth', 'alpha', 'radii',
                 'is_vis', 'capacity', 'last_view', 'report', 'tag', 'need', 'bwd_done')


class StaticRender:
    """See the module docstring.

    ``means3D [P, 3]``, ``opacities [P, 1]``, ``scales [P, 3]``, ``rotations [P, 4]`` and ONE of ``colors_precomp [P, 3]`` /
    ``shs [P, M, 3]``: contiguous float32 tensors on one GPU whose STORAGE stays (update them in place; :meth:`rebind` for new ones).
    ``image_size = (H, W)``; ``capacity``: instances a slot's buffer holds (:func:`required_capacity` measures what a set of
    cameras needs).  ``train``: keep the context a backward needs.  ``stream``: the caller's ``torch.cuda.Stream`` (default: the
    current stream at construction): slot 0 queues its calls there, every further slot owns a stream;
    :meth:`begin` / :meth:`end` / :meth:`join` order them against the caller's, :meth:`slot_stream` hands them out.  ``on_overflow``: ``'repair'`` | ``'raise'`` (module docstring)."""

    GRAD_NAMES = ('means3D', 'means2D', 'opacities', 'scales', 'rotations', 'colors_precomp', 'shs')

    def __init__(self, means3D, opacities, scales, rotations, colors_precomp=None, shs=None, *, image_size, capacity,
                 train=True, stream=None, slots=1, on_overflow='repair'):
        if (colors_precomp is None) == (shs is None):
            raise ValueError('StaticRender: give exactly one of colors_precomp / shs')
        self.device = means3D.device
        if self.device.type != 'cuda':
            raise ValueError('StaticRender needs tensors on a GPU')
        if on_overflow not in ('repair', 'raise'):
            raise ValueError("StaticRender: on_overflow must be 'repair' or 'raise'")
        if int(slots) < 1:
            raise ValueError('StaticRender: slots must be >= 1')
        self.lib = _lib.load()
        self.H, self.W = int(image_size[0]), int(image_size[1])
        self.train = bool(train)
        self.on_overflow = on_overflow
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._pool = _rz._pool()
        self._views = []            # per camera: (settings struct, tensors it points at, image gradients)
        self._jobs = []             # per camera: [(forward job array, backward job array | None) per slot]
        self._outs = []             # registered sets of gradient arrays: (dict, tuple of addresses)
        self._closed = False
        self.forwards = self.repairs = 0
        self._slots = []
        self._set_inputs(means3D, opacities, scales, rotations, colors_precomp, shs)
        cap = (int(capacity) + 63) // 64 * 64
        for s in range(int(slots)):
            sl = _Slot()
            sl.index = s
            # slot 0 queues on the caller's stream, every further slot owns one.  (All slots on streams of their own -- the
            # caller's only coordinating -- was measured: five streams on the four hardware queues a process gets by default
            # cost the independent form 12 %: 10 530 -> 9 230 it/s at S = 4.)
            sl.stream = self.stream if s == 0 else torch.cuda.Stream(device=self.device)
            if sl.stream is not self.stream:
                sl.stream.wait_stream(self.stream)        # (the inputs were written on the caller's stream)
            sl.stream_ptr = ctypes.c_void_p(sl.stream.cuda_stream)
            sl.capacity = cap
            sl.last_view = sl.tag = sl.need = None
            sl.report = self._pool.reserve() if self._pool is not None else None      # (slot, tag, device address): ONE reserved slot
            sl.bwd_done = torch.cuda.Event() if int(slots) > 1 else None
            # (allocated ON the stream the calls are queued on: the caching allocator then hands a freed buffer to nobody before
            #  that stream is past the kernels that used it)
            with torch.cuda.device(self.device), torch.cuda.stream(sl.stream):
                sl.planes = torch.empty((5, self.H, self.W), dtype=_F32, device=self.device)
            sl.color, sl.depth, sl.alpha = sl.planes[:3], sl.planes[3:4], sl.planes[4:5]
            self._slots.append(sl)
            self._alloc_p_sized(sl)
            self._alloc_capacity_sized(sl)

    # ---- storage ------------------------------------------------------------------------------------------------
    def _set_inputs(self, means3D, opacities, scales, rotations, colors_precomp, shs):
        P = int(means3D.shape[0])
        inputs = {'means3D': means3D, 'opacities': opacities, 'scales': scales, 'rotations': rotations,
                  'colors_precomp': colors_precomp, 'shs': shs}
        for name, t in inputs.items():
            if t is not None and not (t.dtype is _F32 and t.is_contiguous() and t.device == self.device and t.shape[0] == P):
                raise ValueError('StaticRender: %s must be a contiguous float32 tensor of P rows on %s' % (name, self.device))
        self.P, self.inputs = P, inputs
        self.sh_M = int(shs.shape[1]) if shs is not None else 0

    def _alloc_p_sized(self, sl):
        sz = _lib.workspace_sizes(self.P, self.W, self.H, 0)
        with torch.cuda.device(self.device), torch.cuda.stream(sl.stream):
            sl.geom = _rz._workspace(sz.geom_bytes, self.device)
            sl.tile = _rz._workspace(sz.tile_bytes, self.device)
            sl.radii = torch.empty((self.P,), dtype=torch.int32, device=self.device)
            sl.is_vis = torch.empty((self.P,), dtype=torch.bool, device=self.device)

    def _alloc_capacity_sized(self, sl):
        sz = _lib.workspace_sizes(self.P, self.W, self.H, sl.capacity)
        with torch.cuda.device(self.device), torch.cuda.stream(sl.stream):
            sl.bin = _rz._workspace(sz.bin_bytes, self.device)
            sl.grad_ws = _rz._workspace(sz.grad_bytes, self.device) if self.train else None

    @property
    def capacity(self):
        return max(sl.capacity for sl in self._slots)

    @property
    def n_slots(self):
        return len(self._slots)

    # slot 0's outputs under the names of a single-slot object
    color = property(lambda self: self._slots[0].color)
    depth = property(lambda self: self._slots[0].depth)
    alpha = property(lambda self: self._slots[0].alpha)
    radii = property(lambda self: self._slots[0].radii)
    is_vis = property(lambda self: self._slots[0].is_vis)

    def outputs(self, slot=0):
        """``{'color', 'depth', 'alpha', 'radii', 'is_vis'}`` of a slot: static tensors its forwards write."""
        sl = self._slots[slot]
        return {'color': sl.color, 'depth': sl.depth, 'alpha': sl.alpha, 'radii': sl.radii, 'is_vis': sl.is_vis}

    # ---- set-up -------------------------------------------------------------------------------------------------
    def add_view(self, raster_settings, dL_dcolor=None, dL_ddepth=None, dL_dalpha=None):
        """Marshal the jobs of one camera (a ``GaussianRasterizationSettings`` whose bg / viewmatrix / projmatrix / campos are
        float32 tensors on this GPU: they are read IN PLACE at every call, e.g. rows of a resident table of views).  The image
        gradients are static tensors too (``dL_dcolor [3, H, W]``; ``None`` for depth / alpha = zero).  Returns the view's index."""
        rs = raster_settings
        if int(rs.image_height) != self.H or int(rs.image_width) != self.W:
            raise ValueError('StaticRender.add_view: the settings are for another image size')
        for name in ('bg', 'viewmatrix', 'projmatrix', 'campos'):
            t = getattr(rs, name)
            if not (isinstance(t, torch.Tensor) and t.device == self.device and t.dtype is _F32 and t.is_contiguous()):
                raise ValueError('StaticRender.add_view: settings.%s must be a contiguous float32 tensor on %s '
                                 '(it is read in place at every call)' % (name, self.device))
        if self.train:
            if dL_dcolor is None:
                raise ValueError('StaticRender.add_view: a training render needs its (static) dL_dcolor tensor')
            for name, g, planes in (('dL_dcolor', dL_dcolor, 3), ('dL_ddepth', dL_ddepth, 1), ('dL_dalpha', dL_dalpha, 1)):
                if g is not None and not (g.dtype is _F32 and g.is_contiguous() and g.device == self.device
                                          and tuple(g.shape) == (planes, self.H, self.W)):
                    raise ValueError('StaticRender.add_view: %s must be a contiguous float32 [%d, H, W] tensor' % (name, planes))
        s = _lib.ExaRasterSettings()
        s.image_height, s.image_width = self.H, self.W
        s.tanfovx, s.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
        s.scale_modifier, s.sh_degree = float(rs.scale_modifier), int(rs.sh_degree)
        s.prefiltered, s.debug = int(bool(rs.prefiltered)), int(bool(rs.debug))
        s.bg, s.viewmatrix, s.projmatrix, s.campos = (rs.bg.data_ptr(), rs.viewmatrix.data_ptr(), rs.projmatrix.data_ptr(),
                                                      rs.campos.data_ptr())
        self._views.append((s, (rs.bg, rs.viewmatrix, rs.projmatrix, rs.campos), (dL_dcolor, dL_ddepth, dL_dalpha)))
        self._jobs.append([self._marshal(len(self._views) - 1, sl) for sl in self._slots])
        return len(self._views) - 1

    def _marshal(self, view, sl):
        """The forward / backward job arrays of camera ``view`` on slot ``sl``."""
        s, _, (dL_dcolor, dL_ddepth, dL_dalpha) = self._views[view]
        f = (_lib.ExaRasterForwardJob * 1)()
        a = f[0]
        a.settings = ctypes.pointer(s)
        a.keep_sorted_keys = 0
        a.host_header, a.header_tag = (sl.report[2], 0) if sl.report is not None else (None, 0)
        base = sl.planes.data_ptr()
        a.out_color, a.out_depth, a.out_alpha = base, base + 12 * self.H * self.W, base + 16 * self.H * self.W
        b = None
        if self.train:
            b = (_lib.ExaRasterBackwardJob * 1)()
            c = b[0]
            c.settings = ctypes.pointer(s)
            c.dL_dcolor, c.dL_ddepth, c.dL_dalpha = dL_dcolor.data_ptr(), _addr(dL_ddepth), _addr(dL_dalpha)
            c.grad_first, c.accumulate, c.used_slots = 0, 0, 0
        self._point(f, b, sl)
        return f, b

    def _point(self, f, b, sl):
        """(Re)write every field of a job pair that names an input, a P-sized or a capacity-sized buffer."""
        i = self.inputs
        a = f[0]
        a.P, a.sh_M = self.P, self.sh_M
        a.means3D, a.shs, a.colors_precomp = _addr(i['means3D']), _addr(i['shs']), _addr(i['colors_precomp'])
        a.opacities, a.scales, a.rotations, a.cov3D_precomp = _addr(i['opacities']), _addr(i['scales']), _addr(i['rotations']), None
        a.radii, a.is_vis = sl.radii.data_ptr(), sl.is_vis.data_ptr()
        a.geom_ws, a.tile_ws, a.bin_ws, a.capacity = sl.geom.data_ptr(), sl.tile.data_ptr(), sl.bin.data_ptr(), sl.capacity
        if b is not None:
            c = b[0]
            c.P, c.sh_M = self.P, self.sh_M
            c.means3D, c.shs, c.colors_precomp = a.means3D, a.shs, a.colors_precomp
            c.opacities, c.scales, c.rotations, c.cov3D_precomp = a.opacities, a.scales, a.rotations, None
            c.radii = a.radii
            c.geom_ws, c.tile_ws, c.bin_ws, c.capacity = a.geom_ws, a.tile_ws, a.bin_ws, sl.capacity
            c.grad_ws = sl.grad_ws.data_ptr()

    def add_grad_outputs(self, **arrays):
        """Register one set of gradient arrays (contiguous float32, P rows: ``means3D [P, 3]``, ``means2D [P, 3]``, ``opacities
        [P, 1]``, ``scales [P, 3]``, ``rotations [P, 4]`` and ``colors_precomp [P, 3]`` or ``shs [P, M, 3]``; a missing name is
        allocated) -- e.g. views into one flat buffer that an all-reduce takes as it is.  Returns the set's index for
        :meth:`backward`; set 0 is created on first use if none was registered."""
        if not self.train:
            raise RuntimeError('StaticRender: not a training render')
        shapes = {'means3D': (self.P, 3), 'means2D': (self.P, 3), 'opacities': (self.P, 1), 'scales': (self.P, 3),
                  'rotations': (self.P, 4)}
        if self.inputs['shs'] is not None:
            shapes['shs'] = (self.P, self.sh_M, 3)
        else:
            shapes['colors_precomp'] = (self.P, 3)
        out = {}
        for name, shape in shapes.items():
            t = arrays.pop(name, None)
            if t is None:
                with torch.cuda.stream(self.stream):
                    t = torch.zeros(shape, dtype=_F32, device=self.device)
            elif not (t.dtype is _F32 and t.is_contiguous() and t.device == self.device and t.numel() == math.prod(shape)):
                raise ValueError('StaticRender.add_grad_outputs: %s must be a contiguous float32 tensor of shape %s' % (name, shape))
            out[name] = t
        if arrays:
            raise ValueError('StaticRender.add_grad_outputs: unknown arrays %s' % sorted(arrays))
        ptrs = (out['means2D'].data_ptr(), out['means3D'].data_ptr(), _addr(out.get('colors_precomp')), out['opacities'].data_ptr(),
                out['scales'].data_ptr(), out['rotations'].data_ptr(), _addr(out.get('shs')))
        self._outs.append((out, ptrs))
        return len(self._outs) - 1

    @property
    def grads(self):
        """Gradient arrays of set 0."""
        if not self._outs:
            self.add_grad_outputs()
        return self._outs[0][0]

    def grad_outputs(self, index):
        return self._outs[index][0]

    def rebind(self, means3D, opacities, scales, rotations, colors_precomp=None, shs=None, capacity=None):
        """New input tensors -- after densification / pruning changed the Gaussian count (``avatar/main/model.py:279-292``), or
        simply other storage.  Waits for the slots' queued work, reallocates what depends on P (splat records, tile
        workspace, ``radii``, ``is_vis``; with ``capacity``: the instance buffers too), keeps cameras, image gradients, images and
        streams, and re-marshals every job.  Registered gradient sets are dropped when P changed (register the new ones)."""
        if (colors_precomp is None) == (shs is None):
            raise ValueError('StaticRender.rebind: give exactly one of colors_precomp / shs')
        if (shs is None) != (self.inputs['shs'] is None):
            raise ValueError('StaticRender.rebind: the colour input (colors_precomp / shs) cannot change kind')
        self._drain()
        old_P = self.P
        self._set_inputs(means3D, opacities, scales, rotations, colors_precomp, shs)
        for sl in self._slots:
            if self.P != old_P:
                self._alloc_p_sized(sl)
            if capacity is not None and (int(capacity) + 63) // 64 * 64 != sl.capacity:
                sl.capacity = (int(capacity) + 63) // 64 * 64
                self._alloc_capacity_sized(sl)
            sl.last_view = sl.tag = sl.need = None
        if self.P != old_P:
            self._outs = []
        for per_slot in self._jobs:
            for sl, (f, b) in zip(self._slots, per_slot):
                self._point(f, b, sl)

    # ---- the calls ----------------------------------------------------------------------------------------------
    def begin(self):
        """Every slot starts behind everything queued on the caller's stream so far (parameters written by an optimizer step,
        image gradients by a loss).  No-op with one slot."""
        for sl in self._slots:
            if sl.stream is not self.stream:
                sl.stream.wait_stream(self.stream)

    def end(self):
        """The caller's stream continues behind everything queued on the slots (their images / gradients may then be read
        there).  No-op with one slot."""
        for sl in self._slots:
            if sl.stream is not self.stream:
                self.stream.wait_stream(sl.stream)

    def join(self, slot):
        """The caller's stream (= slot 0's) continues behind ``slot``'s LAST BACKWARD only -- e.g. the last slot of an accumulation
        chain (``backward(..., after=...)``), where the group's summed gradients are complete.  A caller that wants NO barrier
        between groups coordinates on the slots' own streams instead (:meth:`slot_stream`): whatever consumes the sum is queued on
        the last slot's stream, whatever frees the arrays is waited for on slot 0's (``bench.py --views-in-flight``)."""
        sl = self._slots[slot]
        if sl.bwd_done is not None:
            self.stream.wait_event(sl.bwd_done)

    def slot_stream(self, slot=0):
        """The ``torch.cuda.Stream`` a slot queues its calls on."""
        return self._slots[slot].stream

    def forward(self, view=0, slot=0):
        """Queue the forward of camera ``view`` on ``slot``: images, ``radii``, ``is_vis`` of that slot (:meth:`outputs`)."""
        sl = self._slots[slot]
        f = self._jobs[view][slot][0]
        if sl.report is not None:
            if sl.tag is not None and self.on_overflow == 'raise':
                self._read_report(sl)                   # the slot's previous render: landed long ago unless the host is far ahead
            sl.tag = f[0].header_tag = self._pool._next_tag()
        _lib.check(self.lib.exa_raster_forward_batch(f, 1, 1 if self.train else 0, sl.stream_ptr))
        sl.last_view, sl.need = view, None
        self.forwards += 1
        if self.on_overflow == 'repair':
            need, overflow = self._report(sl)
            if overflow:
                self._grow(sl, need)
                if sl.report is not None:
                    sl.tag = f[0].header_tag = self._pool._next_tag()
                _lib.check(self.lib.exa_raster_forward_batch(f, 1, 1 if self.train else 0, sl.stream_ptr))
                need, overflow = self._report(sl)
                if overflow:
                    raise RuntimeError('exavatar_release_amd.StaticRender: a render overflowed the capacity its own report named')
                self.repairs += 1
            sl.need = need

    def backward(self, out=0, slot=0, accumulate=False, after=None):
        """Queue the backward of ``slot``'s LAST forward (its camera, its static image gradients) into gradient set ``out``.
        ``accumulate``: ADD to what the set's arrays hold (``means2D`` is per render and overwritten) instead of overwriting.
        ``after``: a slot index -- the per-Gaussian kernel of this call (the one that writes the set) runs behind that slot's
        last backward; the blend's backward before it does not wait.  A group of slots that accumulate into one set in a chain
        ``after = previous slot`` produces the sum in that order, bit-identical to the views rendered one after the other."""
        if not self.train:
            raise RuntimeError('StaticRender: not a training render')
        sl = self._slots[slot]
        if sl.last_view is None:
            raise RuntimeError('StaticRender.backward: no forward to differentiate')
        if not self._outs:
            self.add_grad_outputs()
        b = self._jobs[sl.last_view][slot][1]
        c = b[0]
        (c.dL_dmeans2D, c.dL_dmeans3D, c.dL_dcolors, c.dL_dopacity, c.dL_dscales, c.dL_drotations, c.dL_dsh) = self._outs[out][1]
        c.accumulate = 1 if accumulate else 0
        c.used_slots = (sl.need + 63) // 64 if sl.need else 0          # (known in 'repair' mode: one wave per batch slot IN USE)
        if after is None or after == slot:
            _lib.check(self.lib.exa_raster_backward_batch(b, 1, 0, sl.stream_ptr))
        else:
            _lib.check(self.lib.exa_raster_backward_batch(b, 1, _lib.STAGE_BLEND_ONLY, sl.stream_ptr))
            sl.stream.wait_event(self._slots[after].bwd_done)
            _lib.check(self.lib.exa_raster_backward_batch(b, 1, _lib.STAGE_NO_BLEND, sl.stream_ptr))
        if sl.bwd_done is not None:
            sl.bwd_done.record(sl.stream)

    # ---- overflow -----------------------------------------------------------------------------------------------
    def _report(self, sl):
        """(needed instances, overflow flag) of the slot's last forward; waits for it."""
        if sl.report is None:                          # pinned host memory cannot be mapped: read the header back
            hdr = _rz.read_header(sl.tile)
            return hdr[0], hdr[1]
        return _rz._await_report(sl.report[0], sl.tag, sl.stream)

    def _read_report(self, sl):
        need, overflow = self._report(sl)
        sl.tag = None
        if overflow or need > sl.capacity:
            raise RuntimeError('exavatar_release_amd.StaticRender: a render needed %d instances, the buffer holds %d: nothing '
                               'of that render is valid (on_overflow="repair", or a larger capacity)' % (need, sl.capacity))
        sl.need = need

    def _grow(self, sl, need):
        sl.capacity = (int(need * 1.25) + 63) // 64 * 64
        self._alloc_capacity_sized(sl)           # (queued kernels of this slot that use the old buffers: same stream, so the
        #                                           allocator hands the old memory to nobody before they are through)
        for per_slot in self._jobs:
            f, b = per_slot[sl.index]
            f[0].bin_ws, f[0].capacity = sl.bin.data_ptr(), sl.capacity
            if b is not None:
                b[0].bin_ws, b[0].capacity, b[0].grad_ws = sl.bin.data_ptr(), sl.capacity, sl.grad_ws.data_ptr()

    def _drain(self):
        for sl in self._slots:
            sl.stream.synchronize()

    def check(self):
        """Wait for everything queued; ``on_overflow='raise'``: read every slot's outstanding report and raise if a render
        overflowed its buffer."""
        self._drain()
        if self.on_overflow != 'raise':
            return
        err = None
        for sl in self._slots:
            if sl.last_view is None or (sl.report is not None and sl.tag is None):
                continue
            try:
                self._read_report(sl)
            except RuntimeError as e:            # every slot's report is consumed before anything is raised
                err = err or e
        if err is not None:
            raise err

    def close(self):
        """Wait for the device, then drop the buffers and give the report slots back."""
        if self._closed:
            return
        self._closed = True
        try:
            self._drain()
        finally:
            for sl in self._slots:
                if sl.report is not None and self._pool is not None:
                    self._pool.release(sl.report[0])
            self._views, self._jobs, self._outs, self._slots = [], [], [], []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
