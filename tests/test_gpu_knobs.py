"""GPU tests of the boundary knobs the reference never turns (it passes ``scale_modifier=1.0``, ``prefiltered=False``,
``debug=False``: reference avatar/common/nets/module.py:615,620-621) but include/exa_raster.h documents semantics for:

* ``scale_modifier != 1``: images, radii and every gradient against the oracle run with the same modifier;
  ``config.upstream_scale_grad`` on / off = dL/dscale divided by the modifier or not;
* ``prefiltered=True`` is accepted and ignored: results bit-identical to ``False``;
* ``debug=True``: the sync-and-check path of csrc/api.hip (hipStreamSynchronize + error check after every kernel) --
  same results bit for bit, the stream is idle when the call returns, and a launch that cannot be enqueued (a stream
  handle the runtime does not know) comes back as a POSITIVE status (a hipError_t) with ``exa_raster_last_error()``
  naming it -- checked in a child process.

/root/reference is never read here."""
import os
import subprocess
import sys

import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.camera import make_raster_matrices
from exavatar_release_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from oracle import raster_oracle as ro
from tests.helpers import assert_grads_close, assert_image_close, gaussians_near_pixels, rotation_grad_scale

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    from exavatar_release_amd import _lib
    _lib.load()
    exa.config.mode = 'exact'
    exa.config.fixed_capacity = None
    return torch.device('cuda:0')


def _settings(cam, shape, bg, device, **kw):
    tanx, tany, view, proj, campos = make_raster_matrices(cam, shape)
    f = dict(image_height=shape[0], image_width=shape[1], tanfovx=tanx, tanfovy=tany, bg=bg.to(device), scale_modifier=1.0,
             viewmatrix=view.to(device), projmatrix=proj.to(device), sh_degree=0, campos=campos.to(device),
             prefiltered=False, debug=False)
    f.update(kw)
    return GaussianRasterizationSettings(**f)


def _render_gpu(assets, rs, device, G, Gd=None, Ga=None):
    a = {k: assets[k].to(device).requires_grad_(True) for k in NAMES}
    m2 = torch.zeros(a['mean_3d'].shape[0], 3, device=device, requires_grad=True)
    color, radii, depth, alpha = GaussianRasterizer(rs)(
        means3D=a['mean_3d'], means2D=m2, shs=None, colors_precomp=a['rgb'], opacities=a['opacity'], scales=a['scale'],
        rotations=a['rotation'], cov3D_precomp=None)
    loss = (color * G.to(device)).sum()
    if Gd is not None:
        loss = loss + (depth * Gd.to(device)).sum() + (alpha * Ga.to(device)).sum()
    loss.backward()
    return dict(color=color, radii=radii, depth=depth, alpha=alpha, grads={k: a[k].grad for k in NAMES}, m2=m2.grad)


@pytest.mark.parametrize('mod', [0.5, 2.0])
def test_scale_modifier_against_the_oracle(dev, mod):
    """``scale_modifier`` multiplies every scale before the 3D covariance (oracle step 3): radii, lists, images and all
    gradients change with it.  dL/dscale is the derivative with respect to ``scales`` itself (include/exa_raster.h);
    ``config.upstream_scale_grad`` returns upstream's quirk, the same tensor divided by the modifier."""
    H, W, f = 96, 136, 140.0
    assets = scenes.dist_a_random(1800, H, W, seed=31, focal=f)
    assets['scale'][:60] *= 4.0
    cam = scenes.ring_camera(H, W, 3, 11, radius=3.0, center=(0.0, 0.0, 3.0), focal=f)
    g = torch.Generator().manual_seed(7)
    G, Gd, Ga = torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g), torch.randn(1, H, W, generator=g)
    bg = torch.rand(3, generator=g)
    rs = _settings(cam, (H, W), bg, dev, scale_modifier=mod)
    got = _render_gpu(assets, rs, dev, G, Gd, Ga)

    a_cpu = {k: assets[k].clone().requires_grad_(True) for k in NAMES}
    m2 = torch.zeros(a_cpu['mean_3d'].shape[0], 3, requires_grad=True)
    s = ro.settings_from_camera(cam, (H, W), bg)._replace(scale_modifier=mod)
    color, radii, depth, alpha, aux = ro.rasterize(
        means3D=a_cpu['mean_3d'], means2D=m2, shs=None, colors_precomp=a_cpu['rgb'], opacities=a_cpu['opacity'],
        scales=a_cpu['scale'], rotations=a_cpu['rotation'], cov3D_precomp=None, settings=s, return_aux=True)
    ((color * G).sum() + (depth * Gd).sum() + (alpha * Ga).sum()).backward()
    amb = ro.ambiguous_pixel_mask(aux, H, W)
    assert_image_close(got['color'], color, amb, 'img')
    assert_image_close(got['depth'], depth, amb, 'depth')
    assert_image_close(got['alpha'], alpha, amb, 'alpha')
    assert torch.equal(got['radii'].cpu(), radii), 'radii differ'
    near = gaussians_near_pixels(aux['pre'], amb)
    for k in NAMES:
        assert_grads_close(got['grads'][k], a_cpu[k].grad, k, near,
                           abs_scale=rotation_grad_scale(a_cpu['scale'], a_cpu['scale'].grad) if k == 'rotation' else 0.0)
    assert_grads_close(got['m2'], m2.grad, 'mean_2d', near)

    # the modifier really changed the render (the test would pass trivially if it were ignored on both sides)
    plain = _render_gpu(assets, _settings(cam, (H, W), bg, dev), dev, G, Gd, Ga)
    assert not torch.equal(plain['radii'], got['radii'])

    # upstream's quirk: dL/d(modifier * scale) = ours / modifier; everything else untouched
    exa.config.upstream_scale_grad = True
    try:
        up = _render_gpu(assets, rs, dev, G, Gd, Ga)
    finally:
        exa.config.upstream_scale_grad = False
    assert torch.equal(up['grads']['scale'], got['grads']['scale'] / mod)
    for k in ('mean_3d', 'rotation', 'opacity', 'rgb'):
        assert torch.equal(up['grads'][k], got['grads'][k])
    assert torch.equal(up['color'], got['color'])


def test_prefiltered_is_accepted_and_ignored(dev):
    """Upstream uses ``prefiltered`` only to assert that the caller culled already; the library always culls itself
    (include/exa_raster.h): a scene with Gaussians behind the camera renders the same either way, bit for bit."""
    H, W, f = 80, 120, 110.0
    assets = scenes.dist_a_random(1500, H, W, seed=41, focal=f)
    assets['mean_3d'][:100, 2] = -2.0
    cam = scenes.neutral_camera(H, W, focal=f)
    g = torch.Generator().manual_seed(8)
    G, bg = torch.randn(3, H, W, generator=g), torch.rand(3, generator=g)
    a = _render_gpu(assets, _settings(cam, (H, W), bg, dev, prefiltered=False), dev, G)
    b = _render_gpu(assets, _settings(cam, (H, W), bg, dev, prefiltered=True), dev, G)
    for k in ('color', 'radii', 'depth', 'alpha', 'm2'):
        assert torch.equal(a[k], b[k]), k
    for k in NAMES:
        assert torch.equal(a['grads'][k], b['grads'][k]), k
    assert int((a['radii'][:100] > 0).sum()) == 0


def test_debug_mode_synchronises_and_changes_nothing(dev):
    """``debug=True``: hipStreamSynchronize + error check after every kernel (csrc/api.hip debug_sync).  Same bits as the
    asynchronous path, in both instance-buffer modes, and the stream is idle when forward / backward return."""
    H, W = 512, 512
    assets = scenes.dist_b_avatar(60_000, seed=12)
    cam = scenes.ring_camera(H, W, 5, 200, focal=750.0)
    g = torch.Generator().manual_seed(9)
    G, bg = torch.randn(3, H, W, generator=g), torch.rand(3, generator=g)
    ref = _render_gpu(assets, _settings(cam, (H, W), bg, dev), dev, G)
    saved = (exa.config.mode, exa.config.fixed_capacity)
    try:
        for mode in ('exact', 'capacity'):
            exa.config.mode = mode
            rs = _settings(cam, (H, W), bg, dev, debug=True)
            a = {k: assets[k].to(dev).requires_grad_(True) for k in NAMES}
            m2 = torch.zeros(a['mean_3d'].shape[0], 3, device=dev, requires_grad=True)
            torch.cuda.synchronize()
            color, radii, depth, alpha = GaussianRasterizer(rs)(
                means3D=a['mean_3d'], means2D=m2, shs=None, colors_precomp=a['rgb'], opacities=a['opacity'],
                scales=a['scale'], rotations=a['rotation'], cov3D_precomp=None)
            assert torch.cuda.current_stream().query(), 'debug forward returned with work still queued (%s)' % mode
            grads = torch.autograd.grad([color], [a[k] for k in NAMES] + [m2], grad_outputs=[G.to(dev)])
            # (the rasterizer's own kernels are done; autograd may queue its accumulation kernels after them)
            torch.cuda.synchronize()
            assert torch.equal(color, ref['color']) and torch.equal(radii, ref['radii']), mode
            assert torch.equal(depth, ref['depth']) and torch.equal(alpha, ref['alpha']), mode
            for k, gk in zip(NAMES, grads[:5]):
                assert torch.equal(gk, ref['grads'][k]), (mode, k)
            assert torch.equal(grads[5], ref['m2']), mode
    finally:
        exa.config.mode, exa.config.fixed_capacity = saved


_BAD_STREAM_CHILD = r'''
import ctypes, sys, torch
sys.path.insert(0, %(root)r)
from exavatar_release_amd import _lib, scenes
from exavatar_release_amd.camera import make_raster_matrices
lib = _lib.load()
dev = torch.device('cuda:0')
H, W, P = 64, 64, 500
a = scenes.dist_a_random(P, H, W, seed=1, focal=80.0)
cam = scenes.neutral_camera(H, W, focal=80.0)
tanx, tany, view, proj, campos = make_raster_matrices(cam, (H, W))
keep = [t.to(dev).contiguous() for t in (torch.ones(3), view, proj, campos)]
s = _lib.ExaRasterSettings()
s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.scale_modifier = H, W, tanx, tany, 1.0
s.bg, s.viewmatrix, s.projmatrix, s.campos = [t.data_ptr() for t in keep]
s.sh_degree, s.prefiltered, s.debug = 0, 0, int(sys.argv[1])
t = {k: a[k].to(dev).contiguous() for k in ('mean_3d', 'rgb', 'opacity', 'scale', 'rotation')}
sz = _lib.workspace_sizes(P, W, H, 0)
geom = torch.empty(int(sz.geom_bytes), dtype=torch.uint8, device=dev)
tile = torch.empty(int(sz.tile_bytes), dtype=torch.uint8, device=dev)
radii = torch.empty(P, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
p = lambda x: ctypes.c_void_p(x.data_ptr())
bogus = (ctypes.c_uint8 * 4096)()                    # host memory the HIP runtime never handed out as a stream
def call(stream):
    return lib.exa_raster_forward_bin(ctypes.byref(s), P, 0, p(t['mean_3d']), None, p(t['rgb']), p(t['opacity']), p(t['scale']),
                                      p(t['rotation']), None, p(radii), p(geom), p(tile), stream)
rc_ok = call(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
rc_bad = call(ctypes.cast(bogus, ctypes.c_void_p))
err = lib.exa_raster_last_error().decode()
rc_again = call(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print('RESULT', rc_ok, rc_bad, rc_again, '|', err)
'''


@pytest.mark.parametrize('debug', [1, 0])
def test_a_launch_that_cannot_be_enqueued_returns_a_positive_hip_status(dev, debug):
    """Status convention of include/exa_raster.h: 0 ok, < 0 invalid argument, > 0 a hipError_t with
    ``exa_raster_last_error()`` naming the stage.  A stream handle the runtime does not know makes the first launch
    (non-debug) or the first debug synchronisation fail; the library returns that code instead of crashing or
    swallowing it, and the next call on a good stream works (no sticky state in the library).  Runs in a child process:
    a runtime that chose to abort on a foreign handle must not take the test session with it."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, '-c', _BAD_STREAM_CHILD % {'root': ROOT}, str(debug)], capture_output=True, text=True,
                       timeout=240, env=env, cwd=ROOT)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')]
    assert r.returncode == 0 and line, 'child failed: rc %d\n%s\n%s' % (r.returncode, r.stdout[-400:], r.stderr[-800:])
    head, err = line[0].split('|', 1)
    rc_ok, rc_bad, rc_again = [int(v) for v in head.split()[1:4]]
    assert rc_ok == 0 and rc_again == 0
    assert rc_bad > 0, 'expected a hipError_t, got %d (%s)' % (rc_bad, err)
    assert 'HIP error %d' % rc_bad in err
