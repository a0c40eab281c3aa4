"""GPU parity at BASELINE.json's FULL sizes: the HIP path (through the C ABI, via the drop-in Python surface)
against the CPU oracle on the benchmarked workloads themselves -- C2 in both orientations, C3 with 150 k and 167 k
Gaussians on two ring views, C3s (avatar + scene), C5 forward-only with in-kernel SH degree 3 at 2048x2048.
Bars (BASELINE.json north_star): image L-inf 1e-4 on every pixel whose discrete decisions are not within 1e-4 of
a threshold, the number of such ambiguous pixels ASSERTED against a measured budget, radii bit-equal, gradients 1e-3
relative globally AND per Gaussian.  The oracle needs 3-6 s per fwd+bwd view on the GPU box's host cores (~60 s for
the C5 forward, which is why that one and the extra ring views use the C restatement: 0.1-2 s per view).
/root/reference is never read here."""
import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.camera import make_raster_matrices
from oracle import c_oracle as co
from oracle import raster_oracle as ro
from tests.helpers import (assert_grads_close, assert_image_close, gaussians_near_pixels, rotation_grad_scale, grad_stats, image_stats,
                           record_stats)

pytestmark = pytest.mark.gpu

KEYS = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    from exavatar_release_amd import _lib
    _lib.load()          # fail loudly if the HIP library is missing
    exa.config.mode = 'exact'
    exa.config.fixed_capacity = None
    torch.set_num_threads(min(16, torch.get_num_threads()))      # hundreds of OpenMP threads slow the oracle down
    return torch.device('cuda:0')


def _full_parity(tag, assets, shape, cam, dev, max_ambiguous):
    H, W = shape
    g = torch.Generator().manual_seed(1)
    G = torch.randn(3, H, W, generator=g)
    bg = torch.rand(3, generator=g)
    a_gpu = {k: v.to(dev).requires_grad_(True) for k, v in assets.items()}
    out = exa.GaussianRenderer()(a_gpu, shape, {k: v.to(dev) for k, v in cam.items()}, bg.to(dev))
    (out['img'] * G.to(dev)).sum().backward()
    a_cpu = {k: v.clone().requires_grad_(True) for k, v in assets.items()}
    ref = ro.render(a_cpu, shape, cam, bg, return_aux=True)
    (ref['img'] * G).sum().backward()
    amb = ro.ambiguous_pixel_mask(ref['aux'], H, W)
    # measure first, record (gpurun_out/parity_stats.jsonl), then assert
    stats = {'P': assets['mean_3d'].shape[0], 'H': H, 'W': W}
    for name, got, want in (('img', out['img'], ref['img']), ('depth', out['depthmap'], ref['depthmap']),
                            ('alpha', out['mask'], ref['mask'])):
        stats[name] = image_stats(got, want, amb)
    stats['radii_equal'] = bool(torch.equal(out['radius'].cpu(), ref['radius']))
    near = gaussians_near_pixels(ref['aux']['pre'], amb)
    stats['gaussians_near_ambiguous_pixels'] = int(near.sum())
    for k in KEYS:
        stats['grad_' + k] = grad_stats(a_gpu[k].grad, a_cpu[k].grad, near,
                                        rotation_grad_scale(a_cpu['scale'], a_cpu['scale'].grad) if k == 'rotation' else 0.0)
    stats['grad_mean_2d'] = grad_stats(out['mean_2d'].grad, ref['mean_2d'].grad, near)
    record_stats(tag, stats)
    for name in ('img', 'depth', 'alpha'):
        assert_image_close(None, None, amb, name, max_ambiguous, stats=stats[name])
    assert stats['radii_equal'], 'radii differ'
    assert torch.equal(out['is_vis'].cpu(), ref['is_vis'])
    for k in KEYS:
        assert_grads_close(a_gpu[k].grad, a_cpu[k].grad, k, near,
                           abs_scale=rotation_grad_scale(a_cpu['scale'], a_cpu['scale'].grad) if k == 'rotation' else 0.0)
    assert_grads_close(out['mean_2d'].grad, ref['mean_2d'].grad, 'mean_2d', near)
    return stats


# (P, ring view, asserted budget of ambiguous pixels = ~1.3 x the oracle's own count for that view: 378 / 251 / 394 of
# 1 048 576 pixels; C2: 299 / 117; C3s: 1 303; C5: 9 145 of 4.2 M -- profiles/r02_parity.md)
@pytest.mark.parametrize('P,view,budget', [(150_000, 0, 500), (150_000, 37, 350), (167_000, 113, 520)])
def test_c3_full_size_parity_with_oracle(dev, P, view, budget):
    """BASELINE config C3, the benchmarked workload: 150 k (and the real model's ~167 k) avatar-like Gaussians,
    1024x1024, ring views, dense dL/dimage."""
    shape = (1024, 1024)
    _full_parity('c3_P%d_view%d' % (P, view), scenes.dist_b_avatar(P, seed=0), shape,
                 scenes.ring_camera(1024, 1024, view, 200), dev, budget)


@pytest.mark.parametrize('view', [5, 61, 88, 140, 171, 199])
def test_c3_ring_views_against_the_c_oracle(dev, view):
    """The headline workload on six more of the 200 ring views, against the C restatement of the rasterizer
    (oracle/c: 0.1-0.3 s per fwd+bwd view instead of 3-6 s, so breadth is affordable; tests/test_c_oracle.py holds it
    against the PyTorch oracle to 1e-6 incl. the ambiguity margins).  Same bars as above."""
    H = W = 1024
    assets = scenes.dist_b_avatar(150_000, seed=0)
    cam = scenes.ring_camera(H, W, view, 200)
    g = torch.Generator().manual_seed(100 + view)
    G, bg = torch.randn(3, H, W, generator=g), torch.rand(3, generator=g)
    a_gpu = {k: v.to(dev).requires_grad_(True) for k, v in assets.items()}
    out = exa.GaussianRenderer()(a_gpu, (H, W), {k: v.to(dev) for k, v in cam.items()}, bg.to(dev))
    (out['img'] * G.to(dev)).sum().backward()
    ref = co.render(assets, (H, W), cam, bg, dL_dimg=G)
    amb = ref['pixel_margin'] < 1e-4
    stats = {'P': 150_000, 'H': H, 'W': W}
    for name, got, want in (('img', out['img'], ref['img']), ('depth', out['depthmap'], ref['depthmap']),
                            ('alpha', out['mask'], ref['mask'])):
        stats[name] = image_stats(got, want, amb)
        assert_image_close(None, None, amb, name, 600, stats=stats[name])
    stats['radii_equal'] = bool(torch.equal(out['radius'].cpu(), ref['radius']))
    assert stats['radii_equal'], 'radii differ'
    with torch.no_grad():
        so = ro.settings_from_camera(cam, (H, W), bg)
        pre = ro.preprocess(assets['mean_3d'], None, assets['opacity'], assets['scale'], assets['rotation'], None, so,
                            torch.float32)
    near = gaussians_near_pixels(pre, amb)
    scale_ref = ref['grads']['scale']
    for k in KEYS + ('mean_2d',):
        got = a_gpu[k].grad if k != 'mean_2d' else out['mean_2d'].grad
        st = assert_grads_close(got, ref['grads'][k], k, near,
                                abs_scale=rotation_grad_scale(assets['scale'], scale_ref) if k == 'rotation' else 0.0)
        stats['grad_' + k] = st
    record_stats('c3_view%d_vs_c_oracle' % view, stats)


@pytest.mark.parametrize('name,budget', [('c2', 400), ('c2l', 160)])
def test_c2_full_size_parity_with_oracle(dev, name, budget):
    """BASELINE config C2 at full size in both orientations (the survey does not say which): 120 k avatar-like
    Gaussians, 540 x 960 and 960 x 540 -- widths / heights that are no multiple of 16 or 64."""
    assets, shape, cam = scenes.make_config(name)
    _full_parity(name, assets, shape, cam, dev, budget)


def test_c3s_full_size_parity_with_oracle(dev):
    """150 k avatar + 50 k anisotropic, semi-transparent scene Gaussians at 1024x1024 (the scene+human renders of
    avatar/main/model.py:119-167): every sub-tile non-empty, splats spanning hundreds of sub-tiles."""
    assets, shape, cam = scenes.make_config('c3s')
    _full_parity('c3s', assets, shape, cam, dev, 1700)


def test_c5_forward_sh3_full_size_parity_with_oracle(dev):
    """BASELINE config C5 (animation / inference, avatar/main/animate.py:64-66 use case): 300 k Gaussians
    (200 k avatar + 100 k scene), SH degree 3 evaluated in the kernel, 2048x2048, forward only (no context stored)."""
    assets, shape, cam = scenes.make_config('c5')
    H, W = shape
    P = assets['mean_3d'].shape[0]
    sh = scenes.sh_from_rgb(assets['rgb'], 3, seed=5, rest_sigma=0.1)
    g = torch.Generator().manual_seed(2)
    bg = torch.rand(3, generator=g)
    tanx, tany, view, proj, campos = make_raster_matrices(cam, shape)
    st = exa.GaussianRasterizationSettings(H, W, tanx, tany, bg.to(dev), 1.0, view.to(dev), proj.to(dev), 3,
                                           campos.to(dev), False, False)
    with torch.no_grad():
        col, rad, dep, alp = exa.GaussianRasterizer(st)(
            means3D=assets['mean_3d'].to(dev), means2D=torch.zeros(P, 3, device=dev), opacities=assets['opacity'].to(dev),
            shs=sh.to(dev), scales=assets['scale'].to(dev), rotations=assets['rotation'].to(dev))
        so = ro.settings_from_camera(cam, shape, bg, 3)
    # the C restatement (1-2 s instead of 20-60 s for this view; it agrees with the PyTorch oracle on this very workload:
    # radii equal, the 9 145 ambiguous pixels the same but two at the 1e-4 bar, images 4e-7 -- checked on the CPU)
    c = co.rasterize(assets['mean_3d'], assets['opacity'], shs=sh, scales=assets['scale'], rotations=assets['rotation'],
                     settings=so)
    ref = (c['color'], c['radii'], c['depth'], c['alpha'])
    amb = c['pixel_margin'] < 1e-4
    stats = {'P': P, 'H': H, 'W': W}
    for name, got, want in (('img', col, ref[0]), ('depth', dep, ref[2]), ('alpha', alp, ref[3])):
        stats[name] = image_stats(got, want, amb)
    stats['radii_equal'] = bool(torch.equal(rad.cpu(), ref[1]))
    record_stats('c5_fwd_sh3', stats)
    for name in ('img', 'depth', 'alpha'):
        assert_image_close(None, None, amb, name, 12000, stats=stats[name])
    assert stats['radii_equal'], 'radii differ'
