"""GPU parity tests: the HIP path (through the C ABI, via the drop-in Python surface) against the CPU
oracle on the same seeded inputs, against the committed golden fixture, and -- at BASELINE.json's full
sizes -- through size-independent properties.  Tolerances: image 1e-4 L-inf, gradients 1e-3 relative
(BASELINE.json north_star).  /root/reference is never read here."""
import os

import numpy as np
import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.camera import make_raster_matrices
from oracle import raster_oracle as ro
from tests.helpers import (assert_grads_close, assert_image_close, clamped_scene, gaussians_near_pixels,
                           rotation_grad_scale)

pytestmark = pytest.mark.gpu

KEYS = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    from exavatar_release_amd import _lib
    _lib.load()          # fail loudly if the HIP library is missing
    exa.config.mode = 'exact'
    exa.config.fixed_capacity = None
    return torch.device('cuda:0')


def _to(d, dev, grad=True):
    return {k: v.to(dev).requires_grad_(grad) for k, v in d.items()}


def _cmp_render(assets, shape, cam, bg, dev, G, Gd=None, Ga=None):
    H, W = shape
    a_gpu = _to(assets, dev)
    out = exa.GaussianRenderer()(a_gpu, shape, {k: v.to(dev) for k, v in cam.items()}, bg.to(dev))
    loss = (out['img'] * G.to(dev)).sum()
    if Gd is not None:
        loss = loss + (out['depthmap'] * Gd.to(dev)).sum() + (out['mask'] * Ga.to(dev)).sum()
    loss.backward()
    a_cpu = {k: v.clone().requires_grad_(True) for k, v in assets.items()}
    ref = ro.render(a_cpu, shape, cam, bg, return_aux=True)
    lref = (ref['img'] * G).sum()
    if Gd is not None:
        lref = lref + (ref['depthmap'] * Gd).sum() + (ref['mask'] * Ga).sum()
    lref.backward()
    amb = ro.ambiguous_pixel_mask(ref['aux'], H, W)
    assert_image_close(out['img'], ref['img'], amb, 'img')
    assert_image_close(out['depthmap'], ref['depthmap'], amb, 'depth')
    assert_image_close(out['mask'], ref['mask'], amb, 'alpha')
    assert torch.equal(out['radius'].cpu(), ref['radius']), 'radii differ'
    assert torch.equal(out['is_vis'].cpu(), ref['is_vis'])
    near = gaussians_near_pixels(ref['aux']['pre'], amb)
    for k in KEYS:
        assert_grads_close(a_gpu[k].grad, a_cpu[k].grad, k, near,
                           abs_scale=rotation_grad_scale(a_cpu['scale'], a_cpu['scale'].grad) if k == 'rotation' else 0.0)
    assert_grads_close(out['mean_2d'].grad, ref['mean_2d'].grad, 'mean_2d', near)
    return out, ref


def test_c1_full_parity_with_oracle(dev):
    """BASELINE config C1: 10 k random Gaussians, 256x256, colours-precomp, dense dL/dimage."""
    assets, shape, cam = scenes.make_config('c1')
    g = torch.Generator().manual_seed(1)
    G = torch.randn(3, *shape, generator=g)
    bg = torch.rand(3, generator=g)
    _cmp_render(assets, shape, cam, bg, dev, G)


def test_depth_and_alpha_gradients(dev):
    assets = scenes.dist_a_random(3000, 120, 200, seed=9, focal=220.0)
    cam = scenes.neutral_camera(120, 200, focal=220.0)
    g = torch.Generator().manual_seed(2)
    G, Gd, Ga = torch.randn(3, 120, 200, generator=g), torch.randn(1, 120, 200, generator=g), torch.randn(1, 120, 200, generator=g)
    _cmp_render(assets, (120, 200), cam, torch.rand(3, generator=g), dev, G, Gd, Ga)


@pytest.mark.parametrize('shape', [(75, 100), (540 // 4, 960 // 4), (64, 64), (17, 200)])
def test_ragged_image_sizes(dev, shape):
    """Sizes that are not multiples of 8 / 16 / 64 (C2 is 540 wide): border cells, partial sub-tiles."""
    H, W = shape
    f = 1.2 * max(H, W)
    assets = scenes.dist_a_random(2500, H, W, seed=H * 7 + W, focal=f)
    cam = scenes.neutral_camera(H, W, focal=f)
    g = torch.Generator().manual_seed(3)
    _cmp_render(assets, shape, cam, torch.rand(3, generator=g), dev, torch.randn(3, H, W, generator=g))


def test_avatar_like_opaque_isotropic(dev):
    """What HumanGaussian hands over: opacity 1 (0.99 clamp active everywhere), isotropic, identity rotation."""
    H, W = 256, 192
    assets = scenes.dist_b_avatar(20000, seed=4)
    cam = scenes.ring_camera(H, W, 13, 200, focal=300.0)
    g = torch.Generator().manual_seed(4)
    _cmp_render(assets, (H, W), cam, torch.rand(3, generator=g), dev, torch.randn(3, H, W, generator=g))


def test_large_and_offscreen_gaussians(dev):
    """Gaussians spanning many cells, behind the camera, outside the frustum (fov clamp), tiny opacity."""
    H, W = 160, 224
    f = 200.0
    a = scenes.dist_a_random(600, H, W, seed=6, focal=f)
    a['scale'][:40] *= 12.0                      # huge footprints: hundreds of sub-tiles each
    a['mean_3d'][40:60, 2] = -1.0                # behind the camera
    a['mean_3d'][60:80, 0] *= 4.0                # far outside the frustum -> 1.3 tanfov clamp
    a['opacity'][80:120] = 0.002                 # below 1/255: never contributes
    cam = scenes.neutral_camera(H, W, focal=f)
    g = torch.Generator().manual_seed(5)
    out, ref = _cmp_render(a, (H, W), cam, torch.rand(3, generator=g), dev, torch.randn(3, H, W, generator=g))
    assert int((out['radius'][40:60] > 0).sum()) == 0


def test_clamped_gaussians_follow_upstreams_x_grad_mul(dev):
    """Large Gaussians 35-65 % outside the frustum: the +-1.3 tanfov clamp of the EWA projection is active and they
    still reach into the image.  Upstream's backward treats the clamped t.x (t.y) as a constant -- also with respect
    to t.z -- and so do the kernel and both oracles (tests/test_c_oracle.py); plain autograd of the clamp would move
    dL/dmean of such a Gaussian by up to 9 %, far outside the per-Gaussian bar checked here."""
    H, W, f = 96, 128, 150.0
    a = clamped_scene(H, W, f)
    g = torch.Generator().manual_seed(9)
    _cmp_render(a, (H, W), scenes.neutral_camera(H, W, focal=f), torch.rand(3, generator=g), dev,
                torch.randn(3, H, W, generator=g))


@pytest.mark.parametrize('n_deep,o_min', [(1500, 0.004), (2600, 0.008)])
def test_deep_lists(dev, n_deep, o_min):
    """Thousands of faint splats stacked on one 8x8 sub-tile: list lengths beyond the 1024- and 2048-key LDS sorts
    (upstream has no such limit), dozens of 64-splat batches per sub-tile in the per-batch backward, and -- for
    the longer stack -- pixels that stop (T < 1e-4) deep inside the list."""
    H, W = 96, 128        # large enough that the stack's near-threshold pixels stay under the ambiguity budget
    f = 90.0
    g = torch.Generator().manual_seed(n_deep)
    a = scenes.dist_a_random(n_deep + 300, H, W, seed=n_deep, focal=f)
    # centres inside a 6 x 6 px patch in the middle of sub-tile (8, 6); faint, so that nothing saturates early
    depth = 2.0 + 4.0 * torch.rand(n_deep, generator=g)
    a['mean_3d'][:n_deep, 0] = (4.0 + (torch.rand(n_deep, generator=g) - 0.5) * 6.0) / f * depth
    a['mean_3d'][:n_deep, 1] = (4.0 + (torch.rand(n_deep, generator=g) - 0.5) * 6.0) / f * depth
    a['mean_3d'][:n_deep, 2] = depth
    a['scale'][:n_deep] = (0.02 + 0.02 * torch.rand(n_deep, 3, generator=g)) * depth.view(-1, 1)
    a['opacity'][:n_deep] = o_min + 0.004 * torch.rand(n_deep, 1, generator=g)
    cam = scenes.neutral_camera(H, W, focal=f)
    out, ref = _cmp_render(a, (H, W), cam, torch.rand(3, generator=g), dev, torch.randn(3, H, W, generator=g))
    # the stack really is deep: the oracle's 16x16 tile holding that sub-tile lists (almost) all of it
    ranges = ref['aux']['ranges']
    assert int((ranges[:, 1] - ranges[:, 0]).max()) >= n_deep * 0.9
    if n_deep > 2048:
        assert float(ref['aux']['final_T'].min()) < 2e-4          # some pixels ran into the T < 1e-4 stop


def test_equal_depth_ties_are_broken_by_index(dev):
    """Hundreds of splats at EXACTLY the same view depth in one sub-tile list (a fronto-parallel plane): the
    distribution sort finds them all in one bucket and hands the list to the merge sort; the blend order must be the
    ascending Gaussian index, as upstream's stable sort gives it."""
    H, W = 64, 96
    f = 80.0
    g = torch.Generator().manual_seed(91)
    n_plane = 420
    a = scenes.dist_a_random(n_plane + 200, H, W, seed=92, focal=f)
    a['mean_3d'][:n_plane, 2] = 3.0                               # R = I, t = 0: view depth == z, bit for bit
    a['mean_3d'][:n_plane, 0] = (20.0 + 10.0 * torch.rand(n_plane, generator=g) - W / 2 + 0.5) / f * 3.0
    a['mean_3d'][:n_plane, 1] = (12.0 + 10.0 * torch.rand(n_plane, generator=g) - H / 2 + 0.5) / f * 3.0
    a['scale'][:n_plane] = 0.05 + 0.05 * torch.rand(n_plane, 3, generator=g)
    a['opacity'][:n_plane] = 0.02 + 0.05 * torch.rand(n_plane, 1, generator=g)
    cam = scenes.neutral_camera(H, W, focal=f)
    out, ref = _cmp_render(a, (H, W), cam, torch.rand(3, generator=g), dev, torch.randn(3, H, W, generator=g))
    assert int((ref['aux']['pre']['depth'][:n_plane] == 3.0).sum()) == n_plane


def test_empty_and_all_culled(dev):
    H, W = 40, 72
    cam = {k: v.to(dev) for k, v in scenes.neutral_camera(H, W).items()}
    bg = torch.tensor([0.2, 0.4, 0.8], device=dev)
    empty = {'mean_3d': torch.zeros(0, 3), 'scale': torch.zeros(0, 3), 'rotation': torch.zeros(0, 4),
             'opacity': torch.zeros(0, 1), 'rgb': torch.zeros(0, 3)}
    out = exa.GaussianRenderer()(_to(empty, dev, grad=False), (H, W), cam, bg)
    assert torch.equal(out['img'], bg.view(3, 1, 1).expand(3, H, W))
    assert out['radius'].numel() == 0
    a = scenes.dist_a_random(500, H, W, seed=1)
    a['mean_3d'][:, 2] = -2.0
    a_gpu = _to(a, dev)
    out = exa.GaussianRenderer()(a_gpu, (H, W), cam, bg)
    out['img'].sum().backward()
    assert torch.equal(out['img'].detach(), bg.view(3, 1, 1).expand(3, H, W))
    assert float(out["mask"].detach().abs().max()) == 0.0 and int(out["radius"].abs().max()) == 0
    for k in KEYS:
        assert float(a_gpu[k].grad.abs().max()) == 0.0


def test_golden_fixture(dev, golden_dir):
    """Committed golden vector (tests/golden/oracle_small.npz, made by make_golden.py from the oracle)."""
    z = np.load(os.path.join(golden_dir, 'oracle_small.npz'))
    H, W = int(z['H']), int(z['W'])
    a_gpu = {k: torch.tensor(z[k]).to(dev).requires_grad_(True) for k in KEYS}
    cam = {k: v.to(dev) for k, v in scenes.neutral_camera(H, W, focal=float(z['focal'])).items()}
    out = exa.GaussianRenderer()(a_gpu, (H, W), cam, torch.tensor(z['bg']).to(dev))
    loss = (out['img'] * torch.tensor(z['G']).to(dev)).sum() + (out['depthmap'] * torch.tensor(z['Gd']).to(dev)).sum() + \
        (out['mask'] * torch.tensor(z['Ga']).to(dev)).sum()
    loss.backward()
    amb = torch.tensor(z['ambiguous'])
    assert_image_close(out['img'], torch.tensor(z['img']), amb)
    assert_image_close(out['depthmap'], torch.tensor(z['depth']), amb, 'depth')
    assert_image_close(out['mask'], torch.tensor(z['alpha']), amb, 'alpha')
    assert np.array_equal(out['radius'].cpu().numpy(), z['radii'])
    for k in KEYS:
        assert_grads_close(a_gpu[k].grad, torch.tensor(z['grad_' + k]), k)
    assert_grads_close(out['mean_2d'].grad, torch.tensor(z['grad_mean_2d']), 'mean_2d')


def _raster_direct(dev, a, shape, cam, bg, sh_degree=0, **kw):
    tanx, tany, view, proj, campos = make_raster_matrices(cam, shape)
    st = exa.GaussianRasterizationSettings(shape[0], shape[1], tanx, tany, bg.to(dev), 1.0, view.to(dev), proj.to(dev),
                                           sh_degree, campos.to(dev), False, False)
    return exa.GaussianRasterizer(st)(**kw), ro.settings_from_camera(cam, shape, bg, sh_degree)


@pytest.mark.parametrize('deg', [0, 1, 2, 3])
def test_in_kernel_sh_colour(dev, deg):
    """`shs` path: colour = clamp_min(eval_sh + 0.5, 0) with the view-direction gradient."""
    H, W = 96, 128
    f = 150.0
    a = scenes.dist_a_random(1500, H, W, seed=20 + deg, focal=f)
    sh = scenes.sh_from_rgb(a['rgb'], 3, seed=deg, rest_sigma=0.3)      # always 16 coeffs, degree selects how many are used
    cam = scenes.ring_camera(H, W, 3, 40, radius=4.0, center=(0, 0, 4.0), focal=f)
    a['mean_3d'] = a['mean_3d'] + torch.tensor([0.0, 0.0, 0.0])
    g = torch.Generator().manual_seed(7)
    G = torch.randn(3, H, W, generator=g)
    bg = torch.rand(3, generator=g)
    ag = {k: v.to(dev).requires_grad_(True) for k, v in a.items()}
    shg = sh.to(dev).requires_grad_(True)
    m2 = torch.zeros(1500, 3, device=dev, requires_grad=True)
    (col, rad, dep, alp), so = _raster_direct(dev, a, (H, W), cam, bg, deg, means3D=ag['mean_3d'], means2D=m2,
                                              opacities=ag['opacity'], shs=shg, scales=ag['scale'],
                                              rotations=ag['rotation'])
    (col * G.to(dev)).sum().backward()
    ac = {k: v.clone().requires_grad_(True) for k, v in a.items()}
    shc = sh.clone().requires_grad_(True)
    ref = ro.rasterize(ac['mean_3d'], torch.zeros(1500, 3), ac['opacity'], shs=shc, scales=ac['scale'],
                       rotations=ac['rotation'], settings=so, return_aux=True)
    (ref[0] * G).sum().backward()
    amb = ro.ambiguous_pixel_mask(ref[4], H, W)
    assert_image_close(col, ref[0], amb)
    assert_grads_close(shg.grad, shc.grad, 'shs')
    for k in ('mean_3d', 'scale', 'rotation', 'opacity'):
        assert_grads_close(ag[k].grad, ac[k].grad, k)
    # the same render through the plugin surface: assets that carry `sh` instead of `rgb` take the in-kernel path
    a2 = {k: v.detach().clone().requires_grad_(True) for k, v in ag.items() if k != 'rgb'}
    a2['sh'] = shg.detach().clone().requires_grad_(True)
    a2['sh_degree'] = deg
    out = exa.GaussianRenderer()(a2, (H, W), {k: v.to(dev) for k, v in cam.items()}, bg.to(dev))
    (out['img'] * G.to(dev)).sum().backward()
    assert torch.equal(out['img'], col) and torch.equal(out['radius'], rad)
    assert torch.equal(a2['sh'].grad, shg.grad) and torch.equal(a2['mean_3d'].grad, ag['mean_3d'].grad)
    assert torch.equal(out['mean_2d'].grad, m2.grad)


def test_cov3d_precomp_path(dev):
    H, W = 80, 112
    f = 140.0
    a = scenes.dist_a_random(1200, H, W, seed=31, focal=f)
    S = ro.cov3d_from_scale_rot(a['scale'], a['rotation'], 1.0)
    c6 = torch.stack((S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]), 1).contiguous()
    cam = scenes.neutral_camera(H, W, focal=f)
    g = torch.Generator().manual_seed(8)
    G = torch.randn(3, H, W, generator=g)
    bg = torch.rand(3, generator=g)
    c6g = c6.to(dev).requires_grad_(True)
    mg = a['mean_3d'].to(dev).requires_grad_(True)
    m2 = torch.zeros(1200, 3, device=dev, requires_grad=True)
    (col, rad, dep, alp), so = _raster_direct(dev, a, (H, W), cam, bg, 0, means3D=mg, means2D=m2,
                                              opacities=a['opacity'].to(dev), colors_precomp=a['rgb'].to(dev),
                                              cov3D_precomp=c6g)
    (col * G.to(dev)).sum().backward()
    c6c = c6.clone().requires_grad_(True)
    mc = a['mean_3d'].clone().requires_grad_(True)
    ref = ro.rasterize(mc, torch.zeros(1200, 3), a['opacity'], colors_precomp=a['rgb'], cov3D_precomp=c6c,
                       settings=so, return_aux=True)
    (ref[0] * G).sum().backward()
    assert_image_close(col, ref[0], ro.ambiguous_pixel_mask(ref[4], H, W))
    assert_grads_close(c6g.grad, c6c.grad, 'cov3D')
    assert_grads_close(mg.grad, mc.grad, 'mean_3d')


def test_mark_visible(dev):
    a, shape, cam = scenes.make_config('c1')
    a['mean_3d'][:100, 2] = 0.1
    tanx, tany, view, proj, campos = make_raster_matrices(cam, shape)
    st = exa.GaussianRasterizationSettings(shape[0], shape[1], tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev),
                                           proj.to(dev), 0, campos.to(dev), False, False)
    vis = exa.GaussianRasterizer(st).markVisible(a['mean_3d'].to(dev))
    want = ro.mark_visible(a['mean_3d'], ro.settings_from_camera(cam, shape, torch.ones(3)))
    assert torch.equal(vis.cpu(), want)


def _c3_step(dev, assets, shape, view, G, mode='exact'):
    cam = scenes.ring_camera(shape[0], shape[1], view, 200)
    a = _to(assets, dev)
    out = exa.GaussianRenderer()(a, shape, {k: v.to(dev) for k, v in cam.items()}, torch.ones(3, device=dev))
    (out['img'] * G).sum().backward()
    return out, a


def test_c3_full_size_properties(dev):
    """BASELINE config C3 (150 k avatar-like, 1024x1024); oracle parity at this size lives in test_gpu_fullsize.py,
    here the size-independent properties -- determinism (the backward is atomic-free), background
    linearity, alpha == 1 - T, invariance to a permutation of the Gaussians, capacity mode == exact mode."""
    shape = (1024, 1024)
    assets = scenes.dist_b_avatar(150_000, seed=0)
    G = torch.randn(3, *shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    out1, a1 = _c3_step(dev, assets, shape, 37, G)
    out2, a2 = _c3_step(dev, assets, shape, 37, G)
    assert torch.equal(out1['img'], out2['img'])
    for k in KEYS:
        assert torch.equal(a1[k].grad, a2[k].grad), 'backward not deterministic for ' + k
    assert int(out1['is_vis'].sum()) == 150_000
    # background linearity: img(bg1) - img(bg2) == (1 - alpha) (bg1 - bg2)
    cam = {k: v.to(dev) for k, v in scenes.ring_camera(1024, 1024, 37, 200).items()}
    ag = _to(assets, dev, grad=False)
    bg2 = torch.tensor([0.1, 0.5, 0.3], device=dev)
    with torch.no_grad():
        o2 = exa.GaussianRenderer()(ag, shape, cam, bg2)
    T = 1.0 - out1['mask'].detach()
    d = (out1['img'].detach() - o2['img']) - T * (torch.ones(3, device=dev) - bg2).view(3, 1, 1)
    assert float(d.abs().max()) < 2e-6
    assert torch.equal(out1['mask'].detach(), o2['mask'])
    # permutation invariance: per-Gaussian results and the depth order do not depend on the index, except
    # for exact fp32 depth ties, which are broken by index exactly as upstream's stable sort does
    # (150 k depths in [2.7, 3.3] m share ~2.5 M fp32 values: a handful of tied, overlapping pairs exist)
    perm = torch.randperm(150_000, generator=torch.Generator().manual_seed(3))
    ap = {k: v[perm].contiguous() for k, v in assets.items()}
    outp, apg = _c3_step(dev, ap, shape, 37, G)
    dperm = (outp['img'] - out1['img']).detach().abs().amax(0)
    assert int((dperm > 1e-4).sum()) <= 200, 'permutation changed %d pixels' % int((dperm > 1e-4).sum())
    gp, g0 = apg['mean_3d'].grad, a1['mean_3d'].grad[perm.to(dev)]
    assert float((gp - g0).norm() / g0.norm()) < 5e-3
    # capacity mode (no host sync) gives bitwise the same result as exact mode
    exa.config.mode = 'capacity'
    try:
        out3, a3 = _c3_step(dev, assets, shape, 37, G)
        assert torch.equal(out3['img'], out1['img'])
        assert torch.equal(a3['scale'].grad, a1['scale'].grad)
    finally:
        exa.config.mode = 'exact'


@pytest.mark.parametrize('scene', ['c1', 'ragged', 'avatar', 'empty'])
def test_fused_call_equals_two_stage_call_bitwise(dev, scene):
    """The fused forward (capacity / auto mode: one C-ABI call, no host round trip; the scans of the count matrix run
    inside the scatter kernel) against the two-stage protocol (exact mode): images, radii and every gradient bit for bit."""
    if scene == 'c1':
        a, shape, cam = scenes.make_config('c1')
    elif scene == 'ragged':
        shape = (135, 240)
        a = scenes.dist_a_random(2500, shape[0], shape[1], seed=3, focal=300.0)
        cam = scenes.neutral_camera(shape[0], shape[1], focal=300.0)
    elif scene == 'avatar':
        shape = (256, 192)
        a = scenes.dist_b_avatar(20000, seed=4)
        cam = scenes.ring_camera(256, 192, 13, 200, focal=300.0)
    else:
        shape = (40, 72)
        a = scenes.dist_a_random(0, 40, 72, seed=1)
        cam = scenes.neutral_camera(40, 72)
    camd = {k: v.to(dev) for k, v in cam.items()}
    G = torch.randn(3, *shape, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    res = {}
    try:
        for mode in ('exact', 'capacity'):
            exa.config.mode = mode                 # capacity: sized from the exact call of the same shape just before
            ag = _to(a, dev)
            out = exa.GaussianRenderer()(ag, shape, camd, torch.tensor([0.3, 0.2, 0.1], device=dev))
            (out['img'] * G).sum().backward()
            res[mode] = ([out[k].detach().clone() for k in ('img', 'depthmap', 'mask', 'radius')],
                         [ag[k].grad.clone() for k in KEYS] + [out['mean_2d'].grad.clone()])
    finally:
        exa.config.mode = 'exact'
    for x, y in zip(res['exact'][0] + res['exact'][1], res['capacity'][0] + res['capacity'][1]):
        assert torch.equal(x, y)


def test_capacity_overflow_is_reported(dev):
    """``on_overflow = 'raise'``: a capacity-mode render whose buffer is far too small raises from the render call itself
    (the default, 'retry', is covered in tests/test_gpu_edge_cases.py); the device has been left in a sane state."""
    assets, shape, cam = scenes.make_config('c1')
    exa.config.mode = 'capacity'
    exa.config.fixed_capacity = 1000            # far too small
    exa.config.on_overflow = 'raise'
    try:
        a = _to(assets, dev, grad=False)
        with pytest.raises(RuntimeError, match='overflow'), torch.no_grad():
            exa.GaussianRenderer()(a, shape, {k: v.to(dev) for k, v in cam.items()}, torch.ones(3, device=dev))
        torch.cuda.synchronize()
    finally:
        exa.config.mode = 'exact'
        exa.config.fixed_capacity = None
        exa.config.on_overflow = 'retry'


def test_hipgraph_replay_equals_eager(dev):
    """Config 5's mode: forward (+ backward) captured in a hipGraph, replayed with a new camera."""
    H = W = 384
    assets = scenes.dist_b_avatar(30000, seed=2)
    params = [assets[k].to(dev).requires_grad_(True) for k in KEYS]
    mats = [make_raster_matrices(scenes.ring_camera(H, W, k, 12, focal=560.0), (H, W)) for k in range(3)]
    view_s, proj_s, cpos_s = mats[0][2].to(dev).clone(), mats[0][3].to(dev).clone(), mats[0][4].to(dev).clone()
    st = exa.GaussianRasterizationSettings(H, W, mats[0][0], mats[0][1], torch.ones(3, device=dev), 1.0, view_s, proj_s,
                                           0, cpos_s, False, False)
    m2 = torch.zeros(30000, 3, device=dev, requires_grad=True)
    G = torch.randn(3, H, W, device=dev)
    # static output buffers (standard graph hygiene): every step copies its results here and keeps
    # nothing else alive, so no eager-pool block is ever released inside the capture
    out_col = torch.zeros(3, H, W, device=dev)
    out_grads = [torch.zeros_like(p) for p in params]

    def step():
        m3, sc, rot, op, rgb = params
        col, rad, dep, alp = exa.rasterize_gaussians(m3, m2, None, rgb, op, sc, rot, None, st)
        grads = torch.autograd.grad([col], params, grad_outputs=[G])
        # elementwise kernels, not copy_(): a D2D copy_ becomes a hipMemcpyAsync node, and memcpy / memset
        # nodes are what break hipStreamEndCapture in the ROCm runtime bundled with torch 2.10
        torch.add(col.detach(), 0.0, out=out_col)
        for o, g_ in zip(out_grads, grads):
            torch.add(g_, 0.0, out=o)

    def set_view(i):
        view_s.copy_(views_dev[i][0]); proj_s.copy_(views_dev[i][1]); cpos_s.copy_(views_dev[i][2])

    views_dev = [(m[2].to(dev), m[3].to(dev), m[4].to(dev)) for m in mats]
    exa.config.mode = 'exact'
    ref = []
    for i in range(3):
        set_view(i)
        step()
        ref.append((out_col.clone(), [g.clone() for g in out_grads]))
    exa.config.mode = 'capacity'
    exa.config.fixed_capacity = 2_000_000
    try:
        set_view(0)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        for i in (1, 2, 0):
            set_view(i)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(out_col, ref[i][0]), 'graph replay image differs (view %d)' % i
            for g, r in zip(out_grads, ref[i][1]):
                assert torch.equal(g, r)
    finally:
        exa.config.mode = 'exact'
        exa.config.fixed_capacity = None


def test_graphed_renderer_replays_equal_eager_no_grad_renders(dev):
    """GraphedRenderer (BASELINE configs[4]'s mode as a product class: the forward of a fixed-P avatar captured once in a
    hipGraph, replayed per animation frame with new Gaussians and a new camera) against plain no_grad renders through
    GaussianRenderer: images, depth, alpha and radii bit for bit; one capture for a whole turntable; a frame that
    overflows the instance buffer is re-rendered after a re-capture; a new focal length re-captures; SH inputs."""
    H, W, P, f = 160, 192, 4000, 200.0
    base = scenes.dist_a_random(P, H, W, seed=61, focal=f, z_range=(2.5, 5.0))
    g = torch.Generator().manual_seed(62)
    frames = []
    for i in range(4):
        a = {k: v.clone() for k, v in base.items()}
        a['mean_3d'] = a['mean_3d'] + 0.02 * i * torch.randn(P, 3, generator=g)
        a['rgb'] = torch.rand(P, 3, generator=g)
        frames.append(({k: v.to(dev) for k, v in a.items()},
                       {k: t.to(dev) for k, t in scenes.ring_camera(H, W, 3 * i, 40, radius=3.5, center=(0.0, 0.0, 3.5), focal=f).items()},
                       torch.rand(3, generator=g).to(dev)))
    rend = exa.GaussianRenderer()

    def check(gr, a, cam, bg):
        out = gr(a, cam, bg)
        with torch.no_grad():
            ref = rend(a, (H, W), cam, bg)
        for k in ('img', 'depthmap', 'mask', 'radius', 'is_vis'):
            assert torch.equal(out[k], ref[k]), k
    gr = exa.GraphedRenderer(P, (H, W), dev)
    for a, cam, bg in frames:
        check(gr, a, cam, bg)
    assert gr.captures == 1
    # unchanged tensor objects are not copied again, an in-place update is picked up, gr.inputs can be written directly
    a0, cam0, bg0 = frames[0]
    check(gr, a0, cam0, bg0)
    a0['mean_3d'].add_(0.01)
    check(gr, a0, cam0, bg0)
    gr.inputs['rgb'].mul_(0.5)
    check(gr, dict(a0, rgb=gr.inputs['rgb']), cam0, bg0)
    assert gr.captures == 1
    # a new focal length changes tan(fov), which is baked into the kernel arguments: one more capture
    cam2 = dict(frames[0][1]); cam2['focal'] = cam2['focal'] * 1.25
    check(gr, frames[0][0], cam2, frames[0][2])
    assert gr.captures == 2
    # an instance buffer that is too small: the overflowed frame is rendered again after a re-capture
    gr_small = exa.GraphedRenderer(P, (H, W), dev, capacity=64)
    check(gr_small, *frames[1])
    assert gr_small.captures == 2
    check(gr_small, *frames[2])
    # SH inputs evaluated in the kernel
    sh = scenes.sh_from_rgb(base['rgb'], 2, seed=7, rest_sigma=0.2).to(dev)
    gr_sh = exa.GraphedRenderer(P, (H, W), dev, sh_degree=2)
    a, cam, bg = frames[0]
    out = gr_sh({**{k: a[k] for k in ('mean_3d', 'scale', 'rotation', 'opacity')}, 'sh': sh}, cam, bg)
    tanx, tany, view, proj, campos = make_raster_matrices({k: v.cpu() for k, v in cam.items()}, (H, W))
    st = exa.GaussianRasterizationSettings(H, W, tanx, tany, bg, 1.0, view.to(dev), proj.to(dev), 2, campos.to(dev), False, False)
    with torch.no_grad():
        col, rad, dep, alp = exa.GaussianRasterizer(st)(means3D=a['mean_3d'], means2D=torch.zeros(P, 3, device=dev),
                                                        opacities=a['opacity'], shs=sh, scales=a['scale'], rotations=a['rotation'])
    # (the graphed path computes campos = -R^T t on the device, the reference formula inverts the 4x4 view matrix on the
    #  host: equal up to rounding, and campos only enters the SH view direction)
    assert float((out['img'] - col).abs().max()) <= 2e-6 and torch.equal(out['radius'], rad)
    with pytest.raises(ValueError, match='P is fixed'):
        gr({k: v[:-1] for k, v in a.items()}, cam, bg)


def test_fused_densify_stats_match_reference_bookkeeping(dev):
    """one HIP kernel instead of the reference's boolean-mask statements, on a real render's outputs,
    accumulated over three views like a training run does."""
    H, W = 120, 160
    a = scenes.dist_a_random(4000, H, W, seed=31, focal=180.0)
    a['mean_3d'][:500, 2] = -1.0                                   # never visible: statistics must stay untouched
    P = 4000
    acc = torch.rand(P, 1)
    cnt = torch.randint(0, 5, (P, 1)).float()
    rmax = torch.rand(P) * 3
    acc_g, cnt_g, rmax_g = acc.to(dev), cnt.to(dev), rmax.to(dev)
    acc0, cnt0, rmax0 = acc.clone(), cnt.clone(), rmax.clone()
    for v in range(3):
        cam = scenes.ring_camera(H, W, v, 12, radius=3.0, center=(0, 0, 4.0), focal=180.0)
        ag = _to(a, dev)
        out = exa.GaussianRenderer()(ag, (H, W), {k: t.to(dev) for k, t in cam.items()}, torch.ones(3, device=dev))
        out['img'].square().sum().backward()
        g2d, radius = out['mean_2d'].grad, out['radius']
        exa.track_densify_stats(g2d, radius, acc_g, cnt_g, rmax_g)
        acc, cnt, rmax = ro.densify_stats_reference(g2d.cpu(), radius.cpu(), acc, cnt, rmax)
    assert torch.equal(cnt_g.cpu(), cnt) and torch.equal(rmax_g.cpu(), rmax)
    assert torch.allclose(acc_g.cpu(), acc, rtol=1e-6, atol=0)
    assert float(cnt[:500].max()) <= 4.0 and torch.equal(cnt_g.cpu()[:500], cnt[:500])
    # the same statistics riding along in the backward pass itself (ExaRasterBackwardJob.densify_*): no extra kernel
    acc2, cnt2, rmax2 = acc0.to(dev), cnt0.to(dev), rmax0.to(dev)
    for v in range(3):
        cam = scenes.ring_camera(H, W, v, 12, radius=3.0, center=(0, 0, 4.0), focal=180.0)
        ag = _to(a, dev)
        out = exa.GaussianRenderer()(ag, (H, W), {k: t.to(dev) for k, t in cam.items()}, torch.ones(3, device=dev),
                                     densify_stats=(acc2, cnt2, rmax2))
        out['img'].square().sum().backward()
    assert torch.equal(cnt2, cnt_g) and torch.equal(rmax2, rmax_g) and torch.equal(acc2, acc_g)


def test_render_many_is_bit_identical_to_sequential_renders(dev):
    """the five same-camera renders of one ExAvatar iteration (model.py:129-167: scene, human,
    scene+human, human refined, scene+human refined) as five jobs of ONE batched call (one launch per pipeline stage) -- images and every
    gradient equal the sequential result bit for bit."""
    H, W = 128, 160
    f = 170.0
    scene = scenes.dist_a_random(3000, H, W, seed=41, focal=f)
    human = scenes.dist_a_random(1500, H, W, seed=42, focal=f)
    refined = {k: (v + 0.01 * torch.randn(v.shape, generator=torch.Generator().manual_seed(43)) if k == 'mean_3d' else v.clone())
               for k, v in human.items()}
    cam = {k: t.to(dev) for k, t in scenes.neutral_camera(H, W, focal=f).items()}
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    g = torch.Generator().manual_seed(44)
    G = [torch.randn(3, H, W, generator=g).to(dev) for _ in range(5)]

    def run(concurrent):
        s, h, r = _to(scene, dev), _to(human, dev), _to(refined, dev)
        cat = lambda a_, b_: {k: torch.cat((a_[k].detach(), b_[k])) for k in KEYS}      # scene detached, like the reference
        jobs = [(s, (H, W), cam), (h, (H, W), cam, bg), (cat(s, h), (H, W), cam), (r, (H, W), cam, bg), (cat(s, r), (H, W), cam)]
        rend = exa.GaussianRenderer()
        outs = exa.render_many(rend, jobs) if concurrent else [rend(*j) for j in jobs]
        sum((o['img'] * Gi).sum() + o['mask'].sum() for o, Gi in zip(outs, G)).backward()
        torch.cuda.synchronize()
        grads = [t[k].grad.clone() for t in (s, h, r) for k in KEYS] + [o['mean_2d'].grad.clone() for o in outs]
        return [o['img'].detach().clone() for o in outs] + [o['radius'].clone() for o in outs], grads

    imgs_a, grads_a = run(False)
    imgs_b, grads_b = run(True)
    for x, y in zip(imgs_a + grads_a, imgs_b + grads_b):
        assert torch.equal(x, y)


def _frozen_case(dev, nS, nH, H, W, f, seed, big_at=()):
    """scene (constant) + human (trainable) through the constant-prefix path, the plain composite and the oracle."""
    scene = scenes.dist_a_random(nS, H, W, seed=seed, focal=f)
    human = scenes.dist_a_random(nH, H, W, seed=seed + 1, focal=f, z_range=(2.0, 4.0))
    for which, i in big_at:              # splats of >= 64 sub-tiles at chosen indices (wave-cooperative gather)
        t = scene if which == 's' else human
        t['scale'][i] = torch.tensor([0.45, 0.40, 0.35])
        t['mean_3d'][i] = torch.tensor([0.05 * (1 + i % 3), -0.03, 3.0 if which == 's' else 2.9])
        t['opacity'][i] = 0.35
    cam = scenes.neutral_camera(H, W, focal=f)
    cam_d = {k: t.to(dev) for k, t in cam.items()}
    g = torch.Generator().manual_seed(seed + 2)
    G, bg = torch.randn(3, H, W, generator=g), torch.rand(3, generator=g)
    rend = exa.GaussianRenderer()
    # (1) constant-prefix path
    s1, h1 = _to(scene, dev), _to(human, dev)
    o1 = exa.render_many(rend, [(h1, (H, W), cam_d, bg.to(dev), None, s1)])[0]
    ((o1['img'] * G.to(dev)).sum() + o1['mask'].sum()).backward()
    # (2) the reference's formulation: torch.cat((scene.detach(), human)) as ONE Gaussian set
    s2, h2 = _to(scene, dev), _to(human, dev)
    o2 = rend({k: torch.cat((s2[k].detach(), h2[k])) for k in KEYS}, (H, W), cam_d, bg.to(dev))
    ((o2['img'] * G.to(dev)).sum() + o2['mask'].sum()).backward()
    # (3) oracle on the concatenation
    s3 = {k: v.clone() for k, v in scene.items()}
    h3 = {k: v.clone().requires_grad_(True) for k, v in human.items()}
    ref = ro.render({k: torch.cat((s3[k], h3[k])) for k in KEYS}, (H, W), cam, bg, return_aux=True)
    ((ref['img'] * G).sum() + ref['mask'].sum()).backward()
    return (s1, h1, o1), (s2, h2, o2), (h3, ref)


def _assert_same_to_rounding(a, b, name):
    """Two compilations of the same arithmetic: equal up to a few ulp of the tensor's magnitude."""
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 1e-5 * scale + 1e-30, '%s: %g vs scale %g' % (name, float((a - b).abs().max()), scale)


@pytest.mark.parametrize('nS,nH,big_at', [
    (3000, 1500, ()),
    (1000 + 37, 500 + 11, (('s', 1000 + 36), ('h', 0), ('h', 500 + 10))),     # boundary inside a wave / workgroup
    (256, 700, (('h', 3),)),                                                  # prefix = exactly one workgroup
    (5, 64, ()),
])
def test_constant_prefix_matches_concatenation_and_oracle(dev, nS, nH, big_at):
    """ExaRasterBackwardJob.grad_first (the detached scene of ExAvatar's composite renders, model.py:119-126): the
    image equals the plain concatenated render bit for bit, the trainable Gaussians' gradients to rounding (same source,
    but the prefix-aware backward kernels are their own instantiations: multiply-adds may be contracted differently),
    the constants get no gradient, and everything agrees with the oracle."""
    H, W, f = 128, 160, 170.0
    (s1, h1, o1), (s2, h2, o2), (h3, ref) = _frozen_case(dev, nS, nH, H, W, f, 300 + nS % 97, big_at)
    assert torch.equal(o1['img'], o2['img']) and torch.equal(o1['mask'], o2['mask'])
    assert torch.equal(o1['radius'], o2['radius']) and o1['radius'].shape[0] == nS + nH
    assert o1['mean_2d'].shape == (nH, 3)
    assert all(s1[k].grad is None for k in KEYS)
    for k in KEYS:
        _assert_same_to_rounding(h1[k].grad, h2[k].grad, k)
    _assert_same_to_rounding(o1['mean_2d'].grad, o2['mean_2d'].grad[nS:], 'mean_2d')
    amb = ro.ambiguous_pixel_mask(ref['aux'], H, W)
    assert_image_close(o1['img'], ref['img'], amb, 'img')
    assert torch.equal(o1['radius'].cpu(), ref['radius'])
    near = gaussians_near_pixels(ref['aux']['pre'], amb)[nS:]
    for k in KEYS:
        assert_grads_close(h1[k].grad, h3[k].grad, k, near,
                           abs_scale=rotation_grad_scale(h3['scale'], h3['scale'].grad) if k == 'rotation' else 0.0)
    assert_grads_close(o1['mean_2d'].grad, ref['mean_2d'].grad[nS:], 'mean_2d', near)


def test_render_iteration_equals_the_reference_formulation(dev):
    """render_iteration (scene sets shared by the five renders of model.py:119-167, constant scene prefix in the two
    composites) against the reference's own formulation -- five sequential renders with torch.cat((scene.detach(),
    human)): images and radii bit for bit, gradients to rounding (see above)."""
    H, W, f = 128, 160, 170.0
    scene = scenes.dist_a_random(3000, H, W, seed=51, focal=f)
    human = scenes.dist_a_random(1500, H, W, seed=52, focal=f, z_range=(2.0, 4.0))
    refined = {k: (v + 0.01 * torch.randn(v.shape, generator=torch.Generator().manual_seed(53)) if k == 'mean_3d' else v.clone())
               for k, v in human.items()}
    cam = {k: t.to(dev) for k, t in scenes.neutral_camera(H, W, focal=f).items()}
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    g = torch.Generator().manual_seed(54)
    G = [torch.randn(3, H, W, generator=g).to(dev) for _ in range(5)]
    rend = exa.GaussianRenderer()

    def run(shared_sets):
        s, h, r = _to(scene, dev), _to(human, dev), _to(refined, dev)
        if shared_sets:
            res = exa.render_iteration(rend, s, h, r, (H, W), cam, bg)
            outs = [res[k] for k in exa.ITERATION_RENDERS]
        else:
            cat = lambda a_, b_: {k: torch.cat((a_[k].detach(), b_[k])) for k in KEYS}
            jobs = [(s, (H, W), cam), (h, (H, W), cam, bg), (cat(s, h), (H, W), cam), (r, (H, W), cam, bg), (cat(s, r), (H, W), cam)]
            outs = [rend(*j) for j in jobs]
        sum((o['img'] * Gi).sum() + o['mask'].sum() for o, Gi in zip(outs, G)).backward()
        torch.cuda.synchronize()
        grads = [t[k].grad.clone() for t in (s, h, r) for k in KEYS] + [outs[0]['mean_2d'].grad.clone()]
        return [o['img'].detach().clone() for o in outs] + [o['radius'].clone() for o in outs], grads

    imgs_a, grads_a = run(False)
    imgs_b, grads_b = run(True)
    for x, y in zip(imgs_a, imgs_b):
        assert torch.equal(x, y)
    for i, (x, y) in enumerate(zip(grads_a, grads_b)):     # all five jobs of the batched call run the prefix-aware kernels
        _assert_same_to_rounding(y, x, 'grad %d' % i)


def test_last_partial_wave_with_large_gaussian(dev):
    """P % 64 != 0 with a splat of >= 64 sub-tiles at index P - 1 (where densification appends): the per-Gaussian
    backward fetches such a splat's partial sums with the WHOLE wave, so the lanes past P must take part."""
    H, W = 128, 160
    f = 170.0
    P = 1000 + 37
    a = scenes.dist_a_random(P, H, W, seed=77, focal=f)
    for i in (P - 1, P - 2, P - 40):
        a['scale'][i] = torch.tensor([0.45, 0.40, 0.35])       # ~ 60 px radius at z = 3: well over 64 sub-tiles
        a['mean_3d'][i] = torch.tensor([0.02 * (P - i), -0.03, 3.0])
        a['opacity'][i] = 0.35
    cam = scenes.neutral_camera(H, W, focal=f)
    g = torch.Generator().manual_seed(78)
    out, ref = _cmp_render(a, (H, W), cam, torch.rand(3, generator=g), dev, torch.randn(3, H, W, generator=g))
    pre = ref['aux']['pre']
    assert int(pre['tiles_touched'][P - 1]) >= 16              # >= 16 tiles of 16x16 = >= 64 sub-tiles


def test_batched_views_equal_single_renders_and_sum_gradients(dev):
    """exa_raster_*_batch (the view shard of one GPU in one launch per stage): K views of the same Gaussians
    give bit-identical images to K single renders, the shared tensors receive the SUM of the per-view gradients, every
    view keeps its own mean_2d probe -- checked against sequential HIP renders bitwise / to rounding AND against the
    oracle's summed autograd gradients."""
    H, W = 160, 192
    f = 260.0
    K = 3
    assets = scenes.dist_b_avatar(6000, seed=11)
    cams = [scenes.ring_camera(H, W, v, 24, focal=f) for v in (0, 5, 17)]
    g = torch.Generator().manual_seed(12)
    Gs = [torch.randn(3, H, W, generator=g) for _ in range(K)]
    bg = torch.rand(3, generator=g)
    rend = exa.GaussianRenderer()
    to_dev = lambda c: {k: v.to(dev) for k, v in c.items()}
    # sequential single renders
    a_seq = _to(assets, dev)
    outs_seq = [rend(a_seq, (H, W), to_dev(c), bg.to(dev)) for c in cams]
    sum((o['img'] * G.to(dev)).sum() for o, G in zip(outs_seq, Gs)).backward()
    # one batched call
    a_bat = _to(assets, dev)
    outs_bat = exa.render_views(rend, a_bat, (H, W), [to_dev(c) for c in cams], bg.to(dev))
    sum((o['img'] * G.to(dev)).sum() for o, G in zip(outs_bat, Gs)).backward()
    for ob, os_ in zip(outs_bat, outs_seq):
        for k in ('img', 'depthmap', 'mask', 'radius'):
            assert torch.equal(ob[k], os_[k]), k
        assert torch.equal(ob['mean_2d'].grad, os_['mean_2d'].grad)       # per view, not summed
    for k in KEYS:
        gb, gs = a_bat[k].grad, a_seq[k].grad
        ref_mag = float(gs.abs().max()) if k != 'rotation' else rotation_grad_scale(a_seq['scale'], a_seq['scale'].grad)
        assert float((gb - gs).abs().max()) <= 2e-6 * ref_mag, k    # same terms, different summation order
    # oracle: sum of the K views' autograd gradients
    a_cpu = {k: v.clone().requires_grad_(True) for k, v in assets.items()}
    loss = 0
    for c, G in zip(cams, Gs):
        loss = loss + (ro.render(a_cpu, (H, W), c, bg)['img'] * G).sum()
    loss.backward()
    for k in KEYS:
        assert_grads_close(a_bat[k].grad, a_cpu[k].grad, k,
                           abs_scale=rotation_grad_scale(a_cpu['scale'], a_cpu['scale'].grad) if k == 'rotation' else 0.0)


def test_batched_views_with_in_kernel_sh_sum_gradients(dev):
    """The `shs` path through a batch of views of the same Gaussians (the per-Gaussian backward then walks the views in
    one thread and accumulates dL/dsh in place): equals the sum of single renders; P is no multiple of 64."""
    H, W = 96, 128
    f = 150.0
    P = 1500 + 13
    a = scenes.dist_a_random(P, H, W, seed=71, focal=f)
    sh0 = scenes.sh_from_rgb(a['rgb'], 2, seed=3, rest_sigma=0.3)
    g = torch.Generator().manual_seed(72)
    Gs = [torch.randn(3, H, W, generator=g).to(dev) for _ in range(3)]
    bg = torch.rand(3, generator=g).to(dev)
    sts = []
    for v in (0, 7, 19):
        tanx, tany, view, proj, campos = make_raster_matrices(
            scenes.ring_camera(H, W, v, 40, radius=4.0, center=(0, 0, 4.0), focal=f), (H, W))
        sts.append(exa.GaussianRasterizationSettings(H, W, tanx, tany, bg, 1.0, view.to(dev), proj.to(dev), 2, campos.to(dev),
                                                     False, False))

    def leaves():
        t = {k: a[k].to(dev).requires_grad_(True) for k in ('mean_3d', 'scale', 'rotation', 'opacity')}
        t['sh'] = sh0.to(dev).requires_grad_(True)
        return t
    ts, tb = leaves(), leaves()
    kw = lambda t, st: dict(means3D=t['mean_3d'], means2D=torch.zeros(P, 3, device=dev, requires_grad=True), shs=t['sh'],
                            colors_precomp=None, opacities=t['opacity'], scales=t['scale'], rotations=t['rotation'],
                            cov3D_precomp=None, raster_settings=st)
    outs_s = [exa.rasterize_gaussians_batch([kw(ts, st)])[0] for st in sts]
    sum((o[0] * G).sum() for o, G in zip(outs_s, Gs)).backward()
    outs_b = exa.rasterize_gaussians_batch([kw(tb, st) for st in sts])
    sum((o[0] * G).sum() for o, G in zip(outs_b, Gs)).backward()
    for ob, os_ in zip(outs_b, outs_s):
        assert torch.equal(ob[0], os_[0]) and torch.equal(ob[1], os_[1])
    for k in ts:
        gs, gb = ts[k].grad, tb[k].grad
        assert float((gb - gs).abs().max()) <= 3e-6 * float(gs.abs().max()), k


def test_batch_of_heterogeneous_jobs_and_more_than_eight(dev):
    """Jobs with different Gaussian counts AND different image sizes in one batched call, and more jobs than one launch
    holds (groups of eight): every job equals its single render bit for bit, gradients included."""
    specs = [(900, 64, 96), (2500, 120, 80), (64, 40, 40), (1500, 75, 100), (0, 32, 48), (700, 96, 64), (1200, 130, 70),
             (300, 48, 160), (2000, 100, 100), (1100, 88, 56)]
    rend = exa.GaussianRenderer()
    jobs, singles = [], []
    g = torch.Generator().manual_seed(5)
    for i, (P, H, W) in enumerate(specs):
        f = 1.3 * max(H, W)
        a = scenes.dist_a_random(P, H, W, seed=100 + i, focal=f)
        cam = {k: v.to(dev) for k, v in scenes.neutral_camera(H, W, focal=f).items()}
        bg = torch.rand(3, generator=g).to(dev)
        G = torch.randn(3, H, W, generator=g).to(dev)
        jobs.append((_to(a, dev), (H, W), cam, bg, G))
        singles.append((_to(a, dev), (H, W), cam, bg, G))
    outs_s = [rend(*j[:4]) for j in singles]
    sum((o['img'] * j[4]).sum() + o['mask'].sum() for o, j in zip(outs_s, singles)).backward()
    outs_b = exa.render_many(rend, [j[:4] for j in jobs])
    sum((o['img'] * j[4]).sum() + o['mask'].sum() for o, j in zip(outs_b, jobs)).backward()
    for ob, os_, jb, js in zip(outs_b, outs_s, jobs, singles):
        for k in ('img', 'depthmap', 'mask', 'radius'):
            assert torch.equal(ob[k], os_[k]), k
        assert torch.equal(ob['mean_2d'].grad, os_['mean_2d'].grad)
        for k in KEYS:
            assert torch.equal(jb[0][k].grad, js[0][k].grad), k


def test_no_grad_render_matches_training_render(dev):
    """torch.no_grad() renders take the inference variant of the blend kernel (no checkpoints, no masks) although the
    renderer's mean_2d probe requires grad: same image bit for bit."""
    assets, shape, cam = scenes.make_config('c1')
    camd = {k: v.to(dev) for k, v in cam.items()}
    a = _to(assets, dev)
    out = exa.GaussianRenderer()(a, shape, camd, torch.ones(3, device=dev))
    with torch.no_grad():
        out2 = exa.GaussianRenderer()(a, shape, camd, torch.ones(3, device=dev))
    assert torch.equal(out['img'].detach(), out2['img']) and not out2['img'].requires_grad


SSIM_MAP_TOL = 1e-5          # |ssim| <= 1; separable fp32 filtering vs the reference's 2-D conv2d
SSIM_GRAD_TOL = 2e-4         # relative to max |grad|


def test_fused_ssim_and_rgb_loss_match_reference_golden(dev, golden_dir):
    """SURVEY 8f-4, PINNED parity: the fused SSIM kernels and the RGBLoss mirror against outputs and autograd gradients
    of the reference's own class source (tests/golden/ref_ssim.npz), incl. the mask and (clamped) bbox options."""
    z = np.load(os.path.join(golden_dir, 'ref_ssim.npz'))
    t = lambda k: torch.tensor(z[k]).to(dev)
    x, y, mask, bbox, bg, G = t('x'), t('y'), t('mask'), torch.tensor(z['bbox']), t('bg'), t('G')
    ssim, rgb = exa.SSIM(), exa.RGBLoss()
    for name, kw in (('plain', {}), ('mask', {'mask': mask}), ('bbox', {'bbox': bbox})):
        xi = x.clone().requires_grad_(True)
        m = ssim(xi, y, **kw)
        (m * G[:, :, :m.shape[2], :m.shape[3]]).sum().backward()
        ref_m, ref_g = t('ssim_' + name), t('ssim_' + name + '_grad')
        assert float((m.detach() - ref_m).abs().max()) <= SSIM_MAP_TOL, name
        assert float((xi.grad - ref_g).abs().max()) <= SSIM_GRAD_TOL * float(ref_g.abs().max()), name
    for name, kw in (('plain', {}), ('bbox', {'bbox': bbox}), ('maskbg', {'mask': mask, 'bg': bg})):
        xi = x.clone().requires_grad_(True)
        m = rgb(xi, y, **kw)
        (m * G[:, :, :m.shape[2], :m.shape[3]]).sum().backward()
        assert torch.allclose(m.detach(), t('rgb_' + name), rtol=0, atol=1e-7)
        assert torch.equal(xi.grad, t('rgb_' + name + '_grad'))


@pytest.mark.parametrize('shape', [(1, 3, 256, 256), (2, 3, 61, 130), (1, 1, 7, 5)])
def test_fused_ssim_matches_oracle_at_other_sizes(dev, shape):
    """Ragged sizes (tiles cut by the border, images smaller than the 11-tap window) against the pinned oracle."""
    from oracle import loss_oracle as lo
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(*shape, generator=g)
    y = (x + 0.1 * torch.randn(*shape, generator=g)).clamp(0, 1)
    G = torch.randn(*shape, generator=g)
    xc = x.clone().requires_grad_(True)
    mc = lo.ssim_map(xc, y)
    (mc * G).sum().backward()
    xg = x.to(dev).requires_grad_(True)
    mg = exa.SSIM()(xg, y.to(dev))
    (mg * G.to(dev)).sum().backward()
    assert float((mg.detach().cpu() - mc.detach()).abs().max()) <= SSIM_MAP_TOL
    assert float((xg.grad.cpu() - xc.grad).abs().max()) <= SSIM_GRAD_TOL * float(xc.grad.abs().max())


def test_fused_photometric_loss_matches_reference_golden(dev, golden_dir):
    """SURVEY 8f-4, PINNED: the two-kernel L1 + (1 - SSIM) objective and its dL/d(image) against values and autograd
    gradients produced by the reference's own RGBLoss / SSIM classes combined as avatar/main/model.py:197-198 (human
    render, bbox crop) and :214-215 (scene render, 1 - mask) do (tests/golden/make_golden_ssim.py)."""
    z = np.load(os.path.join(golden_dir, 'ref_ssim.npz'))
    t = lambda k: torch.tensor(z[k]).to(dev)
    x, y, mask, bbox = t('x'), t('y'), t('mask'), torch.tensor(z['bbox'])
    crit = exa.PhotometricLoss()
    for name, kw in (('human', {'bbox': bbox}), ('scene', {'l1_weight': 1 - mask, 'ssim_mask': 1 - mask})):
        xi = x.clone().requires_grad_(True)
        L, l1m, ssm = crit(xi, y, return_terms=True, **kw)
        (3.0 * L).backward()                                     # a non-unit upstream gradient
        ref_L, ref_g = float(z['photo_' + name]), t('photo_' + name + '_grad')
        assert abs(float(L) - ref_L) <= 2e-6, (name, float(L), ref_L)
        assert float((xi.grad / 3.0 - ref_g).abs().max()) <= SSIM_GRAD_TOL * float(ref_g.abs().max()), name
        assert abs(0.8 * float(l1m) + 0.2 * (1 - float(ssm)) - float(L)) <= 1e-6


def test_fused_photometric_loss_feeds_the_rasterizer_backward(dev):
    """End to end: render -> PhotometricLoss -> backward; same parameter gradients as the composed-ops loss built from the
    SSIM / RGBLoss drop-ins on the same render (ragged size: tiles cut by the border, bbox crop)."""
    H, W = 150, 200
    f = 240.0
    a = scenes.dist_a_random(3000, H, W, seed=51, focal=f)
    cam = {k: v.to(dev) for k, v in scenes.neutral_camera(H, W, focal=f).items()}
    g = torch.Generator().manual_seed(52)
    target = torch.rand(1, 3, H, W, generator=g).to(dev)
    bbox = torch.tensor([[20.0, -4.0, 150.0, 120.0]])
    rend, crit = exa.GaussianRenderer(), exa.PhotometricLoss()
    grads = []
    for fused in (True, False):
        ag = _to(a, dev)
        img = rend(ag, (H, W), cam, torch.ones(3, device=dev))['img'][None]
        if fused:
            L = crit(img, target, bbox=bbox)
        else:
            L = (exa.RGBLoss()(img, target, bbox=bbox) * 0.8).mean() + ((1 - exa.SSIM()(img, target, bbox=bbox)) * 0.2).mean()
        L.backward()
        grads.append((float(L), {k: ag[k].grad.clone() for k in KEYS}))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6
    for k in KEYS:
        d = (grads[0][1][k] - grads[1][1][k]).abs().max()
        assert float(d) <= 1e-4 * float(grads[1][1][k].abs().max()), k


def test_ssim_is_differentiable_in_the_target_too(dev):
    """The reference's SSIM is plain autograd and differentiates both arguments (loss.py:31-74); the fused kernels get the
    target's gradient from the symmetric call."""
    from oracle import loss_oracle as lo
    g = torch.Generator().manual_seed(61)
    x = torch.rand(1, 3, 45, 70, generator=g)
    y = (x + 0.15 * torch.randn(1, 3, 45, 70, generator=g)).clamp(0, 1)
    G = torch.randn(1, 3, 45, 70, generator=g)
    xc, yc = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    (lo.ssim_map(xc, yc) * G).sum().backward()
    xg, yg = x.to(dev).requires_grad_(True), y.to(dev).requires_grad_(True)
    (exa.SSIM()(xg, yg) * G.to(dev)).sum().backward()
    for got, ref in ((xg.grad, xc.grad), (yg.grad, yc.grad)):
        assert float((got.cpu() - ref).abs().max()) <= SSIM_GRAD_TOL * float(ref.abs().max())


@pytest.mark.gpu
def test_footprint_cull_leaves_the_result_alone(tmp_path):
    """The exact footprint test of the sub-tile binning (csrc/binning.hip) drops (splat, sub-tile) instances whose 64 pixels
    all fail the per-pixel alpha rule, so it may not change any output beyond the rounding of the transmittance products
    (the blend multiplies them in groups of four list entries, and shorter lists group differently).  Same scene -- avatar
    splats plus large anisotropic scene splats -- rendered fwd + bwd by two fresh processes with the test on and off."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ('1', '0'):
        path = str(tmp_path / ('dump_%s.npz' % flag))
        r = subprocess.run([sys.executable, os.path.join(root, 'tests', '_render_dump.py'), path], cwd=root,
                           env=dict(os.environ, EXA_FOOTPRINT=flag), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
    on, off = outs
    assert np.array_equal(on['radius'], off['radius'])
    for k in ('img', 'depth', 'mask'):
        d = np.abs(on[k] - off[k])
        # a T < 1e-4 stop may flip where T (1 - alpha) sits within an ulp of the threshold: budget of 2 pixels
        assert int((d > 2e-6 * max(1.0, float(np.abs(off[k]).max()))).sum()) <= 2 * on[k].shape[0], (k, float(d.max()))
    for k in on.files:
        if 'grad' not in k:
            continue
        scale = float(np.abs(off[k]).max()) + 1e-30
        assert float(np.abs(on[k] - off[k]).max()) <= 2e-5 * scale, (k, float(np.abs(on[k] - off[k]).max()), scale)


def test_one_workgroup_per_cell_binning_gives_the_same_lists(tmp_path):
    """Images of >= 1024 cells (2048 x 2048 px) bin their sub-tiles with ONE workgroup per cell that counts and scatters in
    a single launch (csrc/binning.hip, SINGLE_PART_CELLS) and sort their short lists (<= 64 keys) one wave per list in a
    launch of their own (csrc/render_fwd.hip, SPLIT_SORT_SUBTILES); the C5 full-size test runs through both, forward only.
    Forced on for a small scene through the developer knobs: keys land in the same sub-tile ranges and the depth sort puts
    them in the same order, so every output -- images and all gradients -- must be bit-identical to the default path."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({'EXA_BIN_SINGLE_CELLS': '1', 'EXA_SORT_SPLIT_SUBTILES': '1'}, {}):
        path = str(tmp_path / ('dump_%d.npz' % len(outs)))
        r = subprocess.run([sys.executable, os.path.join(root, 'tests', '_render_dump.py'), path], cwd=root,
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
    single, parts = outs
    for k in parts.files:
        assert np.array_equal(single[k], parts[k]), k
