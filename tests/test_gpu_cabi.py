"""GPU test of the single-render C entry points as a foreign binding would call them (the ctypes stub a
maintainer of the reference would add around ``avatar/common/nets/module.py:632-640``): ``exa_raster_workspace_sizes`` ->
``exa_raster_forward_bin`` -> header read-back -> ``exa_raster_forward_render`` -> ``exa_raster_backward`` (upstream's two-stage
protocol), and ``exa_raster_forward`` with a fixed capacity -> ``exa_raster_backward`` (no host round trip).  No torch autograd,
no Python binding in between: raw pointers, caller-owned workspaces, the stream handle.  Held against the Python binding (which
goes through the ``*_batch`` entry points) bit for bit -- so the two sets of entry points cannot drift apart -- with every output
buffer pre-filled with NaN (``All non-NULL outputs are fully written``) and with optional job fields the single-render calls do
not expose left at their defaults.  /root/reference is never read here."""
import ctypes

import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import _lib, scenes
from exavatar_release_amd.camera import make_raster_matrices

pytestmark = pytest.mark.gpu
NAMES = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    _lib.load()
    return torch.device('cuda:0')


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _nan(shape, dev, dtype=torch.float32):
    return torch.full(shape, float('nan'), dtype=dtype, device=dev)


def _settings(H, W, cam, bg, dev, keep):
    tanx, tany, view, proj, campos = make_raster_matrices(cam, (H, W))
    keep += [t.to(dev).contiguous() for t in (bg, view, proj, campos)]
    s = _lib.ExaRasterSettings()
    s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.scale_modifier = H, W, tanx, tany, 1.0
    s.bg, s.viewmatrix, s.projmatrix, s.campos = [t.data_ptr() for t in keep[-4:]]
    s.sh_degree, s.prefiltered, s.debug = 0, 0, 0
    return s


def _through_python(a, cam, bg, H, W, G, Gd, Ga, dev):
    """The same render through the drop-in surface (GaussianRenderer -> *_batch entry points), exact mode."""
    saved = (exa.config.mode, exa.config.fixed_capacity)
    exa.config.mode, exa.config.fixed_capacity = 'exact', None
    try:
        leaves = {k: a[k].clone().requires_grad_(True) for k in NAMES}
        out = exa.GaussianRenderer()(leaves, (H, W), {k: v.to(dev) for k, v in cam.items()}, bg.to(dev))
        loss = (out['img'] * G).sum() + (out['depthmap'] * Gd).sum() + (out['mask'] * Ga).sum()
        loss.backward()
        torch.cuda.synchronize()
        return out, {k: leaves[k].grad for k in NAMES}, out['mean_2d'].grad
    finally:
        exa.config.mode, exa.config.fixed_capacity = saved


@pytest.mark.parametrize('fused', [False, True])
def test_single_render_entry_points_end_to_end(dev, fused):
    lib = _lib.load()
    H, W, P = 200, 232, 6000                       # neither a multiple of 64 px
    a = {k: v.to(dev).contiguous() for k, v in scenes.dist_a_random(P, H, W, seed=3, focal=260.0).items()}
    cam = scenes.ring_camera(H, W, 3, 40, radius=3.2, center=(0.0, 0.0, 3.0), focal=260.0)
    bg = torch.tensor([0.1, 0.6, 0.3])
    g = torch.Generator().manual_seed(4)
    G, Gd, Ga = [torch.randn(c, H, W, generator=g).to(dev) for c in (3, 1, 1)]
    ref_out, ref_grads, ref_m2 = _through_python(a, cam, bg, H, W, G, Gd, Ga, dev)

    keep = []
    s = _settings(H, W, cam, bg, dev, keep)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sz0 = _lib.workspace_sizes(P, W, H, 0)
    radii = torch.full((P,), -7, dtype=torch.int32, device=dev)
    geom = torch.empty(int(sz0.geom_bytes), dtype=torch.uint8, device=dev)
    tile = torch.empty(int(sz0.tile_bytes), dtype=torch.uint8, device=dev)
    color, depth, alpha = _nan((3, H, W), dev), _nan((1, H, W), dev), _nan((1, H, W), dev)
    inputs = (_p(a['mean_3d']), None, _p(a['rgb']), _p(a['opacity']), _p(a['scale']), _p(a['rotation']), None)
    if fused:
        # a capacity from anywhere (here: twice what the reference render needed); the header says whether it was enough
        capacity = 2 * int(exa.rasterizer._seen_D[(dev.index, P, H, W)])
        bins = torch.empty(int(_lib.workspace_sizes(P, W, H, capacity).bin_bytes), dtype=torch.uint8, device=dev)
        _lib.check(lib.exa_raster_forward(ctypes.byref(s), P, 0, *inputs, _p(radii), _p(geom), _p(tile), _p(bins), capacity,
                                          _p(color), _p(depth), _p(alpha), 1, stream))
        hdr = tile[:16].view(torch.int32).cpu()
        assert int(hdr[1]) == 0 and 0 < int(hdr[0]) <= capacity
    else:
        _lib.check(lib.exa_raster_forward_bin(ctypes.byref(s), P, 0, *inputs, _p(radii), _p(geom), _p(tile), stream))
        hdr = tile[:16].view(torch.int32).cpu()                    # the host round trip upstream makes at this point
        capacity = max(int(hdr[0]), 64)
        bins = torch.empty(int(_lib.workspace_sizes(P, W, H, capacity).bin_bytes), dtype=torch.uint8, device=dev)
        _lib.check(lib.exa_raster_forward_render(ctypes.byref(s), P, _p(geom), _p(tile), _p(bins), capacity,
                                                 _p(color), _p(depth), _p(alpha), 1, stream))
    torch.cuda.synchronize()
    assert torch.equal(color, ref_out['img']) and torch.equal(depth, ref_out['depthmap']) and torch.equal(alpha, ref_out['mask'])
    assert torch.equal(radii, ref_out['radius'])

    grad_ws = torch.empty(int(_lib.workspace_sizes(P, W, H, capacity).grad_bytes), dtype=torch.uint8, device=dev)
    d_m2, d_m3, d_col, d_op = _nan((P, 3), dev), _nan((P, 3), dev), _nan((P, 3), dev), _nan((P, 1), dev)
    d_sc, d_rot = _nan((P, 3), dev), _nan((P, 4), dev)
    _lib.check(lib.exa_raster_backward(ctypes.byref(s), P, 0, *inputs, _p(radii), _p(geom), _p(tile), _p(bins), capacity,
                                       _p(G), _p(Gd), _p(Ga), _p(grad_ws),
                                       _p(d_m2), _p(d_m3), _p(d_col), _p(d_op), _p(d_sc), _p(d_rot), None, None, stream))
    torch.cuda.synchronize()
    got = {'mean_3d': d_m3, 'scale': d_sc, 'rotation': d_rot, 'opacity': d_op, 'rgb': d_col}
    for k in NAMES:
        assert not bool(torch.isnan(got[k]).any()), k
        assert torch.equal(got[k], ref_grads[k]), '%s differs (max %.3e)' % (k, float((got[k] - ref_grads[k]).abs().max()))
    assert torch.equal(d_m2, ref_m2)
    assert float(d_m3.abs().sum()) > 0 and bool((radii == 0).any()) and bool((radii > 0).any())
    # colour gradient only (the ExAvatar training case): NULL depth / alpha gradients, NULL outputs nobody wants
    d_m3b = _nan((P, 3), dev)
    _lib.check(lib.exa_raster_backward(ctypes.byref(s), P, 0, *inputs, _p(radii), _p(geom), _p(tile), _p(bins), capacity,
                                       _p(G), None, None, _p(grad_ws), None, _p(d_m3b), None, None, None, None, None, None, stream))
    torch.cuda.synchronize()
    assert not bool(torch.isnan(d_m3b).any()) and float(d_m3b.abs().sum()) > 0


def test_select_row_steps_through_a_resident_schedule(dev):
    """``exa_raster_select_row``: row (counter mod rows) of a device-resident table -> dst, counter + 1 -- eagerly and as the first
    node of a replayed hipGraph (how bench.py switches views, and how a turntable of precomputed cameras,
    ``avatar/main/animate_view_rot.py:104``, can be rendered without per-frame host work)."""
    lib = _lib.load()
    rows, width = 5, 48
    table = torch.arange(rows * width, dtype=torch.float32, device=dev).reshape(rows, width)
    counter = torch.tensor([3], dtype=torch.int32, device=dev)
    dst = torch.full((width,), -1.0, device=dev)

    def call():
        _lib.check(lib.exa_raster_select_row(_p(table), rows, width, _p(counter), _p(dst),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for want in (3, 4, 0, 1):
        call()
        torch.cuda.synchronize()
        assert torch.equal(dst, table[want]) and int(counter) == (want + 1) % rows
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call()                                              # warm-up outside the capture (row 2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    out = torch.zeros(width, device=dev)
    with torch.cuda.graph(g):
        call()
        torch.mul(dst, 2.0, out=out)                        # a consumer inside the same graph
    for want in (3, 4, 0, 1, 2, 3):
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, 2.0 * table[want]) and int(counter) == (want + 1) % rows
    assert lib.exa_raster_select_row(None, rows, width, _p(counter), _p(dst), None) < 0
    assert lib.exa_raster_select_row(_p(table), 0, width, _p(counter), _p(dst), None) < 0
