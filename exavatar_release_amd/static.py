"""``StaticRender``: one render configuration driven straight through the C ABI (``include/exa_raster.h``) with STATIC
storage -- the caller's input tensors, one set of workspaces, one set of output images and (training) gradient arrays,
all allocated once -- and the job structures of every camera marshalled once.  A forward is ONE ``ctypes`` call
(``exa_raster_forward_batch``: five kernel launches), a backward ONE (``exa_raster_backward_batch``: two launches);
nothing is allocated, converted or filled per call.

Why it exists.  The drop-in autograd surface (``GaussianRasterizer``, reference ``avatar/common/nets/module.py:632-640``)
costs the host ~140 us per forward and ~170 us per backward (autograd node, output tensors, workspaces, 60 ``ctypes``
fields): more than the 145 us the GPU needs for both on the C3 workload, which is why rounds 2-5 replayed it from a
hipGraph.  A graph replay pays for that with ~4 us between two launches of the graph and needs its camera copied into a
static block by one more kernel node (4.6 us).  With the host at ~30 us per step the plain launches are not launch-bound,
the next step's launches queue up behind the running one, and the camera is read where it lies (every view's settings point
into the resident table of views): ``bench.py --launch abi``.  It is also the shape of a native (C++) trainer's inner loop:
what the C ABI is for.

Not autograd: gradients land in the arrays given to (or allocated by) the object; combine with ``torch`` by treating
``.grads`` as the ``.grad`` of the parameters.  Same kernels, same results as the autograd surface bit for bit
(``tests/test_gpu_static.py``).

Overflow of the instance buffer (``capacity``): every forward reports its header into a pinned-host slot; the reports are
read without waiting at the following calls and all of them at :meth:`check` -- an overflowed render raises there (its
images and gradients are not to be used; the autograd surface is the path that repairs an overflow inside the call)."""
import collections
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
        geom = torch.empty(int(sz.geom_bytes), dtype=torch.uint8, device=device)
        tile = torch.empty(int(sz.tile_bytes), dtype=torch.uint8, device=device)
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


class StaticRender:
    """See the module docstring.

    ``means3D [P, 3]``, ``opacities [P, 1]``, ``scales [P, 3]``, ``rotations [P, 4]`` and ONE of ``colors_precomp [P, 3]`` /
    ``shs [P, M, 3]``: contiguous float32 tensors on one GPU whose STORAGE stays (update them in place).
    ``image_size = (H, W)``; ``capacity``: instances the buffer holds (:func:`required_capacity` measures what a set of cameras needs).
    ``train``: keep the context a backward needs.  ``stream``: the ``torch.cuda.Stream`` every call is queued on (default: the
    current stream at construction)."""

    GRAD_NAMES = ('means3D', 'means2D', 'opacities', 'scales', 'rotations', 'colors_precomp', 'shs')

    def __init__(self, means3D, opacities, scales, rotations, colors_precomp=None, shs=None, *, image_size, capacity,
                 train=True, stream=None):
        if (colors_precomp is None) == (shs is None):
            raise ValueError('StaticRender: give exactly one of colors_precomp / shs')
        self.device = means3D.device
        if self.device.type != 'cuda':
            raise ValueError('StaticRender needs tensors on a GPU')
        self.lib = _lib.load()
        self.P = int(means3D.shape[0])
        self.H, self.W = int(image_size[0]), int(image_size[1])
        self.train = bool(train)
        self.inputs = {'means3D': means3D, 'opacities': opacities, 'scales': scales, 'rotations': rotations,
                       'colors_precomp': colors_precomp, 'shs': shs}
        for name, t in self.inputs.items():
            if t is not None and not (t.dtype is _F32 and t.is_contiguous() and t.device == self.device and t.shape[0] == self.P):
                raise ValueError('StaticRender: %s must be a contiguous float32 tensor of P rows on %s' % (name, self.device))
        self.sh_M = int(shs.shape[1]) if shs is not None else 0
        self.capacity = (int(capacity) + 63) // 64 * 64
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._stream_ptr = ctypes.c_void_p(self.stream.cuda_stream)
        sz = _lib.workspace_sizes(self.P, self.W, self.H, self.capacity)
        # (allocated ON the stream the calls are queued on: the caching allocator then hands a freed buffer to nobody before that
        #  stream is past the kernels that used it)
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            self._geom = torch.empty(int(sz.geom_bytes), dtype=torch.uint8, device=self.device)
            self._tile = torch.empty(int(sz.tile_bytes), dtype=torch.uint8, device=self.device)
            self._bin = torch.empty(int(sz.bin_bytes), dtype=torch.uint8, device=self.device)
            self._grad_ws = torch.empty(int(sz.grad_bytes), dtype=torch.uint8, device=self.device) if self.train else None
            planes = torch.empty((5, self.H, self.W), dtype=_F32, device=self.device)
            self.color, self.depth, self.alpha = planes[:3], planes[3:4], planes[4:5]
            self.radii = torch.empty((self.P,), dtype=torch.int32, device=self.device)
            self.is_vis = torch.empty((self.P,), dtype=torch.bool, device=self.device)
        self._planes = planes
        self._views = []            # (settings struct, forward job array, backward job array or None)
        self._keep = []             # tensors the structs point to
        self._outs = []             # registered sets of gradient arrays: (dict, tuple of addresses)
        self._pool = _rz._pool()
        self._pending = collections.deque()      # (slot, tag) of reports not read yet
        self._last_view = None
        self._closed = False
        self.forwards = 0

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
        s = _lib.ExaRasterSettings()
        s.image_height, s.image_width = self.H, self.W
        s.tanfovx, s.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
        s.scale_modifier, s.sh_degree = float(rs.scale_modifier), int(rs.sh_degree)
        s.prefiltered, s.debug = int(bool(rs.prefiltered)), int(bool(rs.debug))
        s.bg, s.viewmatrix, s.projmatrix, s.campos = (rs.bg.data_ptr(), rs.viewmatrix.data_ptr(), rs.projmatrix.data_ptr(),
                                                      rs.campos.data_ptr())
        self._keep += [rs.bg, rs.viewmatrix, rs.projmatrix, rs.campos, dL_dcolor, dL_ddepth, dL_dalpha]
        i = self.inputs
        f = (_lib.ExaRasterForwardJob * 1)()
        a = f[0]
        a.settings = ctypes.pointer(s)
        a.P, a.sh_M = self.P, self.sh_M
        a.means3D, a.shs, a.colors_precomp = _addr(i['means3D']), _addr(i['shs']), _addr(i['colors_precomp'])
        a.opacities, a.scales, a.rotations, a.cov3D_precomp = _addr(i['opacities']), _addr(i['scales']), _addr(i['rotations']), None
        a.radii, a.is_vis = self.radii.data_ptr(), self.is_vis.data_ptr()
        a.geom_ws, a.tile_ws, a.bin_ws, a.capacity = self._geom.data_ptr(), self._tile.data_ptr(), self._bin.data_ptr(), self.capacity
        base = self._planes.data_ptr()
        a.out_color, a.out_depth, a.out_alpha = base, base + 12 * self.H * self.W, base + 16 * self.H * self.W
        a.keep_sorted_keys = 0
        a.host_header, a.header_tag = None, 0
        b = None
        if self.train:
            if dL_dcolor is None:
                raise ValueError('StaticRender.add_view: a training render needs its (static) dL_dcolor tensor')
            for name, g, planes in (('dL_dcolor', dL_dcolor, 3), ('dL_ddepth', dL_ddepth, 1), ('dL_dalpha', dL_dalpha, 1)):
                if g is not None and not (g.dtype is _F32 and g.is_contiguous() and g.device == self.device
                                          and tuple(g.shape) == (planes, self.H, self.W)):
                    raise ValueError('StaticRender.add_view: %s must be a contiguous float32 [%d, H, W] tensor' % (name, planes))
            b = (_lib.ExaRasterBackwardJob * 1)()
            c = b[0]
            c.settings = ctypes.pointer(s)
            c.P, c.sh_M = self.P, self.sh_M
            c.means3D, c.shs, c.colors_precomp = a.means3D, a.shs, a.colors_precomp
            c.opacities, c.scales, c.rotations, c.cov3D_precomp = a.opacities, a.scales, a.rotations, None
            c.radii = a.radii
            c.geom_ws, c.tile_ws, c.bin_ws, c.capacity = a.geom_ws, a.tile_ws, a.bin_ws, self.capacity
            c.dL_dcolor, c.dL_ddepth, c.dL_dalpha = dL_dcolor.data_ptr(), _addr(dL_ddepth), _addr(dL_dalpha)
            c.grad_ws = self._grad_ws.data_ptr()
            c.grad_first, c.accumulate, c.used_slots = 0, 0, 0
        self._views.append((s, f, b))
        return len(self._views) - 1

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

    # ---- the two calls ------------------------------------------------------------------------------------------
    def forward(self, view=0):
        """Queue the forward of camera ``view``: images into ``.color`` / ``.depth`` / ``.alpha``, ``.radii``, ``.is_vis``."""
        s, f, _ = self._views[view]
        pool = self._pool
        if pool is not None:
            pend = self._pending
            while pend and pool.words[4 * pend[0][0] + 3] == pend[0][1]:       # reports that have landed: read, no waiting
                self._read_report(*pend.popleft())
            if len(pend) >= 256:                                                # the host is 256 renders ahead of the device: wait for the oldest
                slot, tag = pend.popleft()
                _rz._await_report(slot, tag, self.stream)
                self._read_report(slot, tag)
            slot, tag, addr = pool.take()
            f[0].host_header, f[0].header_tag = addr, tag
            pend.append((slot, tag))
        _lib.check(self.lib.exa_raster_forward_batch(f, 1, 1 if self.train else 0, self._stream_ptr))
        self._last_view = view
        self.forwards += 1

    def backward(self, out=0):
        """Queue the backward of the LAST forward (its camera, its static image gradients) into gradient set ``out``."""
        if not self.train:
            raise RuntimeError('StaticRender: not a training render')
        if self._last_view is None:
            raise RuntimeError('StaticRender.backward: no forward to differentiate')
        if not self._outs:
            self.add_grad_outputs()
        b = self._views[self._last_view][2]
        c = b[0]
        (c.dL_dmeans2D, c.dL_dmeans3D, c.dL_dcolors, c.dL_dopacity, c.dL_dscales, c.dL_drotations, c.dL_dsh) = self._outs[out][1]
        _lib.check(self.lib.exa_raster_backward_batch(b, 1, 0, self._stream_ptr))

    # ---- overflow -----------------------------------------------------------------------------------------------
    def _read_report(self, slot, tag):
        w = self._pool.words
        need, overflow = int(w[4 * slot]), int(w[4 * slot + 1])
        if overflow or need > self.capacity:
            self._pending.clear()
            raise RuntimeError('exavatar_release_amd.StaticRender: a render needed %d instances, the buffer holds %d: nothing '
                               'of that render is valid (build the object with a larger capacity)' % (need, self.capacity))

    def check(self):
        """Wait for everything queued and read every outstanding report; raises if a render overflowed its buffer."""
        self.stream.synchronize()
        if self._pool is None:
            hdr = _rz.read_header(self._tile)
            if hdr[1] or hdr[0] > self.capacity:
                raise RuntimeError('exavatar_release_amd.StaticRender: the last render needed %d instances, the buffer holds %d'
                                   % (hdr[0], self.capacity))
            return
        while self._pending:
            slot, tag = self._pending.popleft()
            _rz._await_report(slot, tag, self.stream)
            self._read_report(slot, tag)

    def close(self):
        """Wait for the device, then drop the buffers."""
        if self._closed:
            return
        self._closed = True
        try:
            self.stream.synchronize()
        finally:
            self._pending.clear()
            self._views, self._keep, self._outs = [], [], []
            self._geom = self._tile = self._bin = self._grad_ws = self._planes = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
