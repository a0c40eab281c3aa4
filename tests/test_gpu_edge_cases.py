"""GPU parity on the inputs the reference PRODUCES but a synthetic benchmark scene never shows (round-2 verdict,
"HIP-path coverage of degenerate inputs"): the seeded edge-case fuzz of tests/test_c_oracle.py through the HIP path,
the warm-up regime of avatar/main/model.py:92-97 (every human splat clamped to scale <= 1e-3: all at the 0.3 px^2
low-pass floor, radius 2) at the full C3 size, scene-like inputs (opacities from a sigmoid down to the 1/255 bar,
unnormalised quaternions as avatar/common/nets/module.py:253-272 can hand them over), the transparent retry after an
instance-buffer overflow, densification statistics shared by the views of one batch, and a two-rank run whose
all-reduced flat gradient must equal the single-process sum over the same views.
Same bars as everywhere: image 1e-4 L-inf off ambiguous pixels, radii bit-equal, gradients 1e-3 relative.
/root/reference is never read here."""
import json
import os
import subprocess
import sys
import warnings

import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import rasterizer as rz
from exavatar_release_amd import scenes
from oracle import c_oracle as co
from oracle import raster_oracle as ro
from tests.helpers import (IMG_TOL, assert_grads_close, assert_image_close, fuzz_case, gaussians_near_pixels, grad_stats,
                           image_stats, record_stats, rotation_grad_scale)

pytestmark = pytest.mark.gpu

KEYS = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    from exavatar_release_amd import _lib
    _lib.load()          # fail loudly if the HIP library is missing
    exa.config.mode = 'exact'
    exa.config.fixed_capacity = None
    torch.set_num_threads(min(16, torch.get_num_threads()))
    return torch.device('cuda:0')


def _to(d, dev, grad=True):
    return {k: v.to(dev).requires_grad_(grad) for k, v in d.items()}


# EXA_FUZZ_TRIALS=n widens the seeded fuzz (rounds 5 and 6 ran it once with 300 seeds: profiles/r06_fuzz_300.log)
@pytest.mark.parametrize('trial', range(int(os.environ.get('EXA_FUZZ_TRIALS', '16'))))
def test_edge_case_fuzz_through_the_hip_path(dev, trial):
    """The 16 seeded trials of tests/test_c_oracle.py::test_edge_case_fuzz_* through GaussianRenderer: opacity 0 / 1 / at
    the 1/255 bar, scales x1e-4 .. x300, centres at / around / behind the near plane and ON the camera plane, unnormalised
    quaternions, ragged image sizes, all three image gradients.  Against the C oracle (values) with the float64 run of
    the PyTorch oracle as arbiter for the gradients, exactly as the CPU fuzz holds the two oracles against each other."""
    a, H, W, cam, G, Gd, Ga, bg = fuzz_case(trial)
    ag = _to(a, dev)
    out = exa.GaussianRenderer()(ag, (H, W), {k: v.to(dev) for k, v in cam.items()}, bg.to(dev))
    ((out['img'] * G.to(dev)).sum() + (out['depthmap'] * Gd.to(dev)).sum() + (out['mask'] * Ga.to(dev)).sum()).backward()
    c = co.render(a, (H, W), cam, bg, dL_dimg=G, dL_ddepth=Gd, dL_dalpha=Ga)
    t = {k: v.clone().requires_grad_(True) for k, v in a.items()}
    r = ro.render(t, (H, W), cam, bg, return_aux=True)
    amb = ro.ambiguous_pixel_mask(r['aux'], H, W) | (c['pixel_margin'] < 1e-4)
    # structure: radii / visibility bit-equal, everything finite, culled Gaussians get exactly zero
    assert torch.equal(out['radius'].cpu(), c['radius']), 'radii differ'
    assert torch.equal(out['is_vis'].cpu(), c['radius'] > 0)
    for k in ('img', 'depthmap', 'mask'):
        assert bool(torch.isfinite(out[k]).all()), k
    culled = c['radius'] == 0
    grads = {k: ag[k].grad.cpu() for k in KEYS}
    grads['mean_2d'] = out['mean_2d'].grad.cpu()
    for k, g in grads.items():
        assert bool(torch.isfinite(g).all()), k
        assert not bool(g[culled].any()), 'culled Gaussians must get zero gradient: ' + k
    # images on every pixel whose decisions are not at a threshold (depth scaled by its magnitude: z reaches 8)
    if bool((~amb).any()):
        for k, ref, scale in (('img', c['img'], 1.0), ('mask', c['mask'], 1.0),
                              ('depthmap', c['depthmap'], 1.0 + float(c['depthmap'].abs().max()))):
            d = (out[k].detach().cpu() - ref).abs()
            d = d.amax(0) if d.dim() == 3 else d
            assert float(d[~amb].max()) <= IMG_TOL * scale, '%s off by %.3e' % (k, float(d[~amb].max()))
    # gradients: float64 arbiter, only when no decision flips between float32 and float64 and no pixel is ambiguous
    # (tiny images where EVERY Gaussian covers every pixel: one flipped decision moves every gradient)
    t64 = {k: v.clone().double().requires_grad_(True) for k, v in a.items()}
    r64 = ro.render(t64, (H, W), cam, bg, dtype=torch.float64, return_aux=True)
    ((r64['img'] * G.double()).sum() + (r64['depthmap'] * Gd.double()).sum() + (r64['mask'] * Ga.double()).sum()).backward()
    same = bool((c['n_contrib'] == r['aux']['n_contrib']).all()) and bool((r64['aux']['n_contrib'] == r['aux']['n_contrib']).all()) \
        and bool((r64['radius'] == r['radius']).all())
    # ... or when the HIP image agrees with the oracle on the ambiguous pixels too: then no decision flipped there either
    agree_everywhere = all(float((out[k].detach().cpu() - ref).abs().max()) <= IMG_TOL * sc for k, ref, sc in (
        ('img', c['img'], 1.0), ('mask', c['mask'], 1.0), ('depthmap', c['depthmap'], 1.0 + float(c['depthmap'].abs().max()))))
    stats = {'trial': trial, 'H': H, 'W': W, 'P': int(a['mean_3d'].shape[0]), 'n_ambiguous': int(amb.sum()), 'same': same,
             'agree_everywhere': agree_everywhere}
    if same and (agree_everywhere or not bool(amb.any())):
        for k in KEYS + ('mean_2d',):
            ref = (t64[k].grad if k != 'mean_2d' else r64['mean_2d'].grad).float()
            scale = float(ref.abs().max()) + 1e-12
            if k == 'rotation':
                scale = max(scale, float(t64['scale'].grad.abs().max() * t64['scale'].detach().abs().max()))
            err = float((grads[k] - ref).abs().max()) / scale
            stats['grad_' + k] = err
            assert err <= 1e-3, 'grad %s: %.3e of the max-norm' % (k, err)
    record_stats('fuzz_%d' % trial, stats)


def _warmup_assets(P, seed=0):
    """What HumanGaussian hands the renderer during warm-up: Dist-B with `scale.clamp(max=1e-3)`
    (reference avatar/main/model.py:92-97)."""
    a = scenes.dist_b_avatar(P, seed=seed)
    a['scale'] = a['scale'].clamp(max=1e-3)
    return a


def test_warmup_scale_clamp_small(dev):
    """Warm-up regime at a size the PyTorch oracle handles in a second: every splat at the 0.3 px^2 low-pass floor."""
    H, W, f = 192, 160, 260.0
    a = _warmup_assets(12000, seed=3)
    cam = scenes.ring_camera(H, W, 4, 24, focal=f)
    g = torch.Generator().manual_seed(21)
    G, bg = torch.randn(3, H, W, generator=g), torch.rand(3, generator=g)
    ag = _to(a, dev)
    out = exa.GaussianRenderer()(ag, (H, W), {k: v.to(dev) for k, v in cam.items()}, bg.to(dev))
    (out['img'] * G.to(dev)).sum().backward()
    t = {k: v.clone().requires_grad_(True) for k, v in a.items()}
    ref = ro.render(t, (H, W), cam, bg, return_aux=True)
    (ref['img'] * G).sum().backward()
    amb = ro.ambiguous_pixel_mask(ref['aux'], H, W)
    assert int(ref['radius'][ref['radius'] > 0].max()) <= 3        # the regime: radius 2 (3 at the frustum edge)
    for k, w in (('img', ref['img']), ('depthmap', ref['depthmap']), ('mask', ref['mask'])):
        assert_image_close(out[k], w, amb, k)
    assert torch.equal(out['radius'].cpu(), ref['radius'])
    near = gaussians_near_pixels(ref['aux']['pre'], amb)
    for k in KEYS:
        assert_grads_close(ag[k].grad, t[k].grad, k, near,
                           abs_scale=rotation_grad_scale(t['scale'], t['scale'].grad) if k == 'rotation' else 0.0)
    assert_grads_close(out['mean_2d'].grad, ref['mean_2d'].grad, 'mean_2d', near)


@pytest.mark.parametrize('P,view', [(150_000, 0), (167_000, 77)])
def test_warmup_scale_clamp_regime_full_size(dev, P, view):
    """The warm-up regime at the headline size: 150 k (167 k) avatar-like Gaussians with scale <= 1e-3 at 1024x1024 --
    every splat covers ~2x2 sub-tiles with the minimum footprint, lists are short and uniform (the opposite corner of
    the binning / sort / blend code from the benchmark scene).  Against the C oracle, same bars as C3."""
    H = W = 1024
    a = _warmup_assets(P)
    cam = scenes.ring_camera(H, W, view, 200)
    g = torch.Generator().manual_seed(300 + view)
    G, bg = torch.randn(3, H, W, generator=g), torch.rand(3, generator=g)
    ag = _to(a, dev)
    out = exa.GaussianRenderer()(ag, (H, W), {k: v.to(dev) for k, v in cam.items()}, bg.to(dev))
    (out['img'] * G.to(dev)).sum().backward()
    ref = co.render(a, (H, W), cam, bg, dL_dimg=G)
    amb = ref['pixel_margin'] < 1e-4
    stats = {'P': P, 'H': H, 'W': W, 'max_radius': int(ref['radius'].max())}
    assert stats['max_radius'] <= 3
    for name, got, want in (('img', out['img'], ref['img']), ('depth', out['depthmap'], ref['depthmap']),
                            ('alpha', out['mask'], ref['mask'])):
        stats[name] = image_stats(got, want, amb)
        assert_image_close(None, None, amb, name, 1500, stats=stats[name])
    assert torch.equal(out['radius'].cpu(), ref['radius']), 'radii differ'
    with torch.no_grad():
        so = ro.settings_from_camera(cam, (H, W), bg)
        pre = ro.preprocess(a['mean_3d'], None, a['opacity'], a['scale'], a['rotation'], None, so, torch.float32)
    near = gaussians_near_pixels(pre, amb)
    for k in KEYS + ('mean_2d',):
        got = ag[k].grad if k != 'mean_2d' else out['mean_2d'].grad
        stats['grad_' + k] = assert_grads_close(
            got, ref['grads'][k], k, near,
            abs_scale=rotation_grad_scale(a['scale'], ref['grads']['scale']) if k == 'rotation' else 0.0)
    record_stats('warmup_P%d_view%d' % (P, view), stats)


def test_scene_like_inputs_sigmoid_opacities_and_raw_quaternions(dev):
    """What SceneGaussian.forward produces (module.py:253-272): opacity = sigmoid(raw) spread down to and below the 1/255
    bar, scale = exp(raw) over two decades, rotation from a 6D parametrisation (here: quaternions left unnormalised, norm
    0.5 .. 2 -- the rasterizer must use them as they are)."""
    H, W, f = 144, 208, 210.0
    P = 5000
    a = scenes.dist_a_random(P, H, W, seed=41, focal=f)
    g = torch.Generator().manual_seed(42)
    a['opacity'] = torch.sigmoid(torch.randn(P, 1, generator=g) * 2.5 - 2.0)         # many below / around 1/255
    a['opacity'][:50] = torch.tensor([1 / 255.0, 0.00392, 0.00393, 0.0040, 0.0])[torch.randint(0, 5, (50,), generator=g)].view(-1, 1)
    a['scale'] = torch.exp(torch.randn(P, 3, generator=g) * 1.0 - 4.0)
    a['rotation'] = a['rotation'] * (0.5 + 1.5 * torch.rand(P, 1, generator=g))
    cam = scenes.ring_camera(H, W, 2, 9, radius=3.0, center=(0.0, 0.0, 3.0), focal=f)
    G, Gd, Ga = torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g), torch.randn(1, H, W, generator=g)
    bg = torch.rand(3, generator=g)
    ag = _to(a, dev)
    out = exa.GaussianRenderer()(ag, (H, W), {k: v.to(dev) for k, v in cam.items()}, bg.to(dev))
    ((out['img'] * G.to(dev)).sum() + (out['depthmap'] * Gd.to(dev)).sum() + (out['mask'] * Ga.to(dev)).sum()).backward()
    t = {k: v.clone().requires_grad_(True) for k, v in a.items()}
    ref = ro.render(t, (H, W), cam, bg, return_aux=True)
    ((ref['img'] * G).sum() + (ref['depthmap'] * Gd).sum() + (ref['mask'] * Ga).sum()).backward()
    amb = ro.ambiguous_pixel_mask(ref['aux'], H, W)
    for k in ('img', 'depthmap', 'mask'):
        assert_image_close(out[k], ref[k], amb, k)
    assert torch.equal(out['radius'].cpu(), ref['radius'])
    near = gaussians_near_pixels(ref['aux']['pre'], amb)
    for k in KEYS:
        assert_grads_close(ag[k].grad, t[k].grad, k, near,
                           abs_scale=rotation_grad_scale(t['scale'], t['scale'].grad) if k == 'rotation' else 0.0)
    assert_grads_close(out['mean_2d'].grad, ref['mean_2d'].grad, 'mean_2d', near)


# ---- sub-tile binning: the adaptive split of the cells over its workgroups ---------------------------------------------
@pytest.mark.parametrize('case', ['few_cells_many_entries', 'one_dense_cell'])
def test_binning_split_extremes(dev, case):
    """csrc/binning.hip write_part_table: a cell gets one workgroup per max(1024, entries / (3 cells)) entries, at most 16,
    out of a fixed grid of 4 per cell.  (a) four cells holding > 20 k entries: the entries-per-part bound is what keeps
    the sum inside the grid; (b) one 64 x 64-px cell holding nearly all of 20 k entries (clamped to 16 parts of more than
    one loop trip each) between empty and nearly empty cells.  Against the PyTorch oracle, same bars as everywhere."""
    H, W, f, P, a = _split_case(case)
    g = torch.Generator().manual_seed(7)
    cam = {'R': torch.eye(3), 't': torch.zeros(3), 'focal': torch.tensor([f, f]), 'princpt': torch.tensor([W / 2.0, H / 2.0])}
    G = torch.randn(3, H, W, generator=g)
    bg = torch.rand(3, generator=g)
    ag = _to(a, dev)
    out = exa.GaussianRenderer()(ag, (H, W), {k: v.to(dev) for k, v in cam.items()}, bg.to(dev))
    (out['img'] * G.to(dev)).sum().backward()
    t = {k: v.clone().requires_grad_(True) for k, v in a.items()}
    ref = ro.render(t, (H, W), cam, bg, return_aux=True)
    (ref['img'] * G).sum().backward()
    amb = ro.ambiguous_pixel_mask(ref['aux'], H, W)
    for k in ('img', 'depthmap', 'mask'):
        assert_image_close(out[k], ref[k], amb, k)
    assert torch.equal(out['radius'].cpu(), ref['radius'])
    near = gaussians_near_pixels(ref['aux']['pre'], amb)
    for k in KEYS:
        assert_grads_close(ag[k].grad, t[k].grad, k, near,
                           abs_scale=rotation_grad_scale(t['scale'], t['scale'].grad) if k == 'rotation' else 0.0)
    assert_grads_close(out['mean_2d'].grad, ref['mean_2d'].grad, 'mean_2d', near)


def _split_case(case):
    g = torch.Generator().manual_seed(11)
    if case == 'few_cells_many_entries':
        H, W, f, P = 128, 128, 190.0, 20_000
        a = scenes.dist_a_random(P, H, W, seed=51, focal=f)
        a['scale'] = a['scale'] * 0.25
    else:
        H, W, f, P = 192, 256, 260.0, 20_000
        a = scenes.dist_a_random(P, H, W, seed=52, focal=f)
        a['scale'] = a['scale'] * 0.2
        m = a['mean_3d']
        dense = torch.arange(P) >= 300                         # all but 300 splats projected into the cell at (1, 1)
        u = 64.0 + 8.0 + 48.0 * torch.rand(P, generator=g)
        v = 64.0 + 8.0 + 48.0 * torch.rand(P, generator=g)
        z = m[:, 2]
        m[dense, 0] = ((u - W / 2.0) / f * z)[dense]
        m[dense, 1] = ((v - H / 2.0) / f * z)[dense]
    a['opacity'] = a['opacity'] * 0.25                         # long lists: no early saturation
    return H, W, f, P, a


# ---- overflow: transparent retry ---------------------------------------------------------------------------------
def _reset_config():
    exa.config.mode = 'exact'
    exa.config.fixed_capacity = None
    exa.config.on_overflow = 'retry'


@pytest.mark.parametrize('capacity', [64, 1024, -64])          # (-64: 64 instances short of what the render needs)
def test_overflow_is_repaired_inside_forward(dev, capacity):
    """Capacity mode with a buffer that is too small -- by far, or by one batch slot: the header report is polled at the end of
    the render's own forward and the render is repaired there -- the image the caller gets, a loss computed from it and the
    gradients are those of the exact-mode render bit for bit, without any warning (nobody saw incomplete outputs);
    ``on_overflow = 'raise'`` raises from the forward call.  Training and ``no_grad`` renders alike."""
    assets, shape, cam = scenes.make_config('c1')
    camd = {k: v.to(dev) for k, v in cam.items()}
    G = torch.randn(3, *shape, generator=torch.Generator().manual_seed(5)).to(dev)
    a_ref = _to(assets, dev)
    ref = exa.GaussianRenderer()(a_ref, shape, camd, torch.ones(3, device=dev))
    loss_ref = ((ref['img'] - G) ** 2).mean()          # a loss whose gradient depends on the image itself
    loss_ref.backward()
    need = rz._seen_D[(dev.index or 0, assets['mean_3d'].shape[0], shape[0], shape[1])]
    cap = capacity if capacity > 0 else need + capacity
    try:
        exa.config.mode, exa.config.fixed_capacity = 'capacity', cap
        n0 = len(rz.overflow_events)
        a = _to(assets, dev)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            out = exa.GaussianRenderer()(a, shape, camd, torch.ones(3, device=dev))
            assert torch.equal(out['img'].detach(), ref['img'].detach())      # complete when the call returns
            assert torch.equal(out['radius'], ref['radius'])
            loss = ((out['img'] - G) ** 2).mean()
            loss.backward()
        assert not [x for x in w if issubclass(x.category, RuntimeWarning)]
        assert len(rz.overflow_events) == n0 + 1 and rz.overflow_events[-1][1:] == (need, cap, 'retried')
        assert float(loss) == float(loss_ref)
        for k in KEYS:
            assert torch.equal(a[k].grad, a_ref[k].grad), k
        assert torch.equal(out['mean_2d'].grad, ref['mean_2d'].grad)
        with torch.no_grad():                                                  # also without autograd
            out = exa.GaussianRenderer()(a, shape, camd, torch.ones(3, device=dev))
        assert torch.equal(out['img'], ref['img'].detach()) and torch.equal(out['depthmap'], ref['depthmap'].detach())
        exa.config.on_overflow = 'raise'
        with pytest.raises(RuntimeError, match='overflow'):
            exa.GaussianRenderer()(_to(assets, dev), shape, camd, torch.ones(3, device=dev))
        with pytest.raises(RuntimeError, match='overflow'), torch.no_grad():
            exa.GaussianRenderer()(_to(assets, dev, grad=False), shape, camd, torch.ones(3, device=dev))
        assert rz.overflow_events[-1][3] == 'raised'
        torch.cuda.synchronize()
    finally:
        _reset_config()


def test_overflow_of_one_job_of_a_batch_is_repaired(dev):
    """A batched call (three views of one model, one launch per stage) whose SECOND job alone overflows: that job is
    re-rendered by itself, the batch's images and the summed gradients equal three exact-mode renders bit for bit."""
    H, W, f, P = 160, 192, 220.0, 6000
    assets = scenes.dist_a_random(P, H, W, seed=51, focal=f)
    cams = [{k: v.to(dev) for k, v in scenes.ring_camera(H, W, k, 8, focal=f).items()} for k in range(3)]
    G = torch.randn(3, H, W, generator=torch.Generator().manual_seed(52)).to(dev)
    a_ref = _to(assets, dev)
    refs = exa.render_views(exa.GaussianRenderer(), a_ref, (H, W), cams)
    sum((o['img'] * G).sum() for o in refs).backward()
    need = rz._seen_D[(dev.index or 0, P, H, W)]
    try:
        exa.config.mode, exa.config.fixed_capacity = 'capacity', [2 * need, 64, 2 * need]
        n0 = len(rz.overflow_events)
        a = _to(assets, dev)
        outs = exa.render_views(exa.GaussianRenderer(), a, (H, W), cams)
        assert len(rz.overflow_events) == n0 + 1 and rz.overflow_events[-1][3] == 'retried'
        for o, r in zip(outs, refs):
            assert torch.equal(o['img'].detach(), r['img'].detach())
        sum((o['img'] * G).sum() for o in outs).backward()
        for k in KEYS:
            assert torch.equal(a[k].grad, a_ref[k].grad), k
        for o, r in zip(outs, refs):
            assert torch.equal(o['mean_2d'].grad, r['mean_2d'].grad)
    finally:
        _reset_config()


def test_two_sets_of_equal_size_alternate_without_errors(dev):
    """Two Gaussian sets with the SAME P (so they share the capacity memo keyed on (P, H, W)) whose instance counts differ
    by > 3x, rendered alternately in 'auto' mode: the first render of the dense set overflows the memo of the sparse one,
    is repaired inside its forward, and from then on the memo covers both -- no RuntimeError, gradients always those of
    the exact-mode render."""
    H, W, f, P = 160, 192, 220.0, 6000
    sparse = scenes.dist_a_random(P, H, W, seed=51, focal=f)
    dense = {k: v.clone() for k, v in sparse.items()}
    dense['scale'] = dense['scale'] * 4.0                     # ~16x the footprint area
    cam = {k: v.to(dev) for k, v in scenes.neutral_camera(H, W, focal=f).items()}
    G = torch.randn(3, H, W, generator=torch.Generator().manual_seed(52)).to(dev)
    bg = torch.ones(3, device=dev)
    refs = []
    for s in (sparse, dense):
        a = _to(s, dev)
        o = exa.GaussianRenderer()(a, (H, W), cam, bg)
        (o['img'] * G).sum().backward()
        refs.append(({k: a[k].grad.clone() for k in KEYS}, o['img'].detach().clone()))
    try:
        exa.config.mode = 'auto'
        rz._seen_D.clear()
        n0 = len(rz.overflow_events)
        with warnings.catch_warnings(record=True):
            warnings.simplefilter('always')
            for it in range(6):
                for s, (gref, iref) in zip((sparse, dense), refs):
                    a = _to(s, dev)
                    o = exa.GaussianRenderer()(a, (H, W), cam, bg)
                    (o['img'] * G).sum().backward()
                    torch.cuda.synchronize()
                    assert torch.equal(o['img'].detach(), iref)
                    for k in KEYS:
                        assert torch.equal(a[k].grad, gref[k]), (it, k)
        assert len(rz.overflow_events) - n0 <= 1              # at most the one retry
    finally:
        _reset_config()


# ---- densification statistics shared by the views of a batch -----------------------------------------------------
def test_views_of_one_batch_may_share_one_set_of_densify_stats(dev):
    """K views of the same Gaussians accumulating into ONE (xyz_grad_accum, track_cnt, radius_max): the batched backward
    sums the K views' statistics in the kernel and writes once per Gaussian (round-2 advisor finding: the per-view
    read-modify-write raced between the waves that split the views).  Equals K sequential renders."""
    H, W, f, P, K = 128, 160, 200.0, 5000, 5
    a = scenes.dist_a_random(P, H, W, seed=61, focal=f)
    cams = [{k: v.to(dev) for k, v in scenes.ring_camera(H, W, v, 11, radius=3.0, center=(0, 0, 4.0), focal=f).items()} for v in range(K)]
    bg = torch.ones(3, device=dev)
    rend = exa.GaussianRenderer()
    seq = [torch.zeros(P, device=dev) for _ in range(3)]
    for c in cams:
        ag = _to(a, dev)
        o = rend(ag, (H, W), c, bg, densify_stats=tuple(seq))
        o['img'].square().sum().backward()
    bat = [torch.zeros(P, device=dev) for _ in range(3)]
    ag = _to(a, dev)
    outs = exa.render_many(rend, [(ag, (H, W), c, bg, tuple(bat)) for c in cams])
    sum(o['img'].square().sum() for o in outs).backward()
    torch.cuda.synchronize()
    assert torch.equal(bat[1], seq[1]) and torch.equal(bat[2], seq[2])
    assert torch.allclose(bat[0], seq[0], rtol=1e-5, atol=1e-12)           # K-term sums in a different order
    assert float(bat[1].max()) == K
    # two DIFFERENT Gaussian sets updating the same statistics in one batch: rejected (would race)
    other = _to(scenes.dist_a_random(P, H, W, seed=62, focal=f), dev)
    with pytest.raises(ValueError, match='densify_stats'):
        exa.render_many(rend, [(ag, (H, W), cams[0], bg, tuple(bat)), (other, (H, W), cams[1], bg, tuple(bat))])


def test_non_contiguous_inputs_in_a_batch_of_views(dev):
    """Round-2 advisor finding: K views passing the same NON-contiguous tensor objects (rgb = feats[:, :3]) used to get K
    private .contiguous() copies and fail the sum_shared pointer check in backward."""
    H, W, f, P = 96, 128, 150.0, 3000
    a = scenes.dist_a_random(P, H, W, seed=63, focal=f)
    feats = torch.cat((a['rgb'], torch.zeros(P, 5)), 1).to(dev).requires_grad_(True)
    ag = _to({k: v for k, v in a.items() if k != 'rgb'}, dev)
    ag['rgb'] = feats[:, :3]
    cams = [{k: v.to(dev) for k, v in scenes.ring_camera(H, W, v, 7, radius=3.0, center=(0, 0, 3.0), focal=f).items()} for v in range(3)]
    outs = exa.render_views(exa.GaussianRenderer(), ag, (H, W), cams, torch.ones(3, device=dev))
    sum(o['img'].sum() for o in outs).backward()
    ref = _to(a, dev)
    loss = 0
    for c in cams:
        loss = loss + exa.GaussianRenderer()(ref, (H, W), c, torch.ones(3, device=dev))['img'].sum()
    loss.backward()
    scale = float(ref['rgb'].grad.abs().max())
    assert float((feats.grad[:, :3] - ref['rgb'].grad).abs().max()) <= 2e-6 * scale
    assert not bool(feats.grad[:, 3:].any())


# ---- two ranks: the all-reduced flat gradient equals the single-process sum over the same views ----------------------
def test_two_rank_reduced_gradient_equals_single_process_sum(tmp_path):
    """Two ranks (gloo, sharing this GPU -- RCCL refuses duplicate devices) rasterize their shards of 6 ring views fwd+bwd,
    pack the gradients with the product's FlatGradAllReducer and all-reduce them; rank 0 then renders the union of the views
    in ONE process (render_views, sum_shared) and compares: <= 1e-6 relative per tensor.  tests/_dist_grad_worker.py."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(29650 + os.getpid() % 200), os.path.join(ROOT, 'tests', '_dist_grad_worker.py')]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    res = json.loads(line)
    assert res['world'] == 2 and res['views'] == 6
    for k, v in res['rel_err'].items():
        assert v <= 1e-6, (k, v)
    record_stats('two_rank_gradient', res)


# ---- composite renders: merged lists instead of binning the concatenation -------------------------------------------
def _iteration_case(dev, scene, human, refined, H, W, f, bg, seed, cam=None):
    """render_iteration three ways: merge (composites from the sources' sorted lists), constant prefix (round 2) and the
    reference's own formulation (five renders of torch.cat((scene.detach(), human)))."""
    cam = cam or scenes.neutral_camera(H, W, focal=f)
    camd = {k: t.to(dev) for k, t in cam.items()}
    g = torch.Generator().manual_seed(seed)
    G = [torch.randn(3, H, W, generator=g).to(dev) for _ in range(5)]
    Gd = [torch.randn(1, H, W, generator=g).to(dev) for _ in range(5)]
    rend = exa.GaussianRenderer()
    res = {}
    for how in ('merge', 'prefix', 'reference'):
        s, h, r = _to(scene, dev), _to(human, dev), _to(refined, dev)
        if how == 'reference':
            cat = lambda a_, b_: {k: torch.cat((a_[k].detach(), b_[k])) for k in KEYS}      # noqa: E731
            jobs = [(s, (H, W), camd), (h, (H, W), camd, bg), (cat(s, h), (H, W), camd), (r, (H, W), camd, bg), (cat(s, r), (H, W), camd)]
            outs = [rend(*j) for j in jobs]
        else:
            out = exa.render_iteration(rend, s, h, r, (H, W), camd, bg, merge=(how == 'merge'))
            outs = [out[k] for k in exa.ITERATION_RENDERS]
        sum((o['img'] * Gi).sum() + (o['depthmap'] * Gdi).sum() + o['mask'].sum() for o, Gi, Gdi in zip(outs, G, Gd)).backward()
        torch.cuda.synchronize()
        m2 = [o['mean_2d'].grad.clone() for o in outs]
        if how == 'reference':                       # the composites' probes cover cat(scene, human): keep the human rows
            nS = scene['mean_3d'].shape[0]
            m2[2], m2[4] = m2[2][nS:], m2[4][nS:]
        res[how] = dict(imgs=[o[k].detach().clone() for o in outs for k in ('img', 'depthmap', 'mask')],
                        radii=[o['radius'].clone() for o in outs], m2=m2,
                        grads=[t[k].grad.clone() for t in (s, h, r) for k in KEYS])
    return res


def _assert_iteration_agrees(res):
    ref = res['reference']
    for how in ('merge', 'prefix'):
        got = res[how]
        for i, (x, y) in enumerate(zip(got['imgs'], ref['imgs'])):
            assert torch.equal(x, y), '%s image %d' % (how, i)
        for x, y in zip(got['radii'], ref['radii']):
            assert torch.equal(x, y)
        for i, (x, y) in enumerate(zip(got['grads'] + got['m2'], ref['grads'] + ref['m2'])):
            scale = float(y.abs().max())
            assert float((x - y).abs().max()) <= 1e-5 * scale + 1e-30, '%s grad %d: %g of %g' % (how, i, float((x - y).abs().max()), scale)


def test_composite_renders_merge_the_sorted_lists_of_their_sources(dev):
    """exa_raster_forward_compose_batch (csrc/compose.hip): scene + human without binning the concatenation -- the composite
    reuses the sources' splat records and merges their sorted per-sub-tile lists.  Images, depth, alpha and radii equal the
    reference's five renders (torch.cat((scene.detach(), human))) bit for bit, every gradient to rounding."""
    H, W, f = 128, 160, 170.0
    scene = scenes.dist_a_random(3000, H, W, seed=51, focal=f)
    human = scenes.dist_a_random(1500, H, W, seed=52, focal=f, z_range=(2.0, 4.0))
    refined = {k: (v + 0.01 * torch.randn(v.shape, generator=torch.Generator().manual_seed(53)) if k == 'mean_3d' else v.clone())
               for k, v in human.items()}
    res = _iteration_case(dev, scene, human, refined, H, W, f, torch.tensor([0.1, 0.2, 0.3], device=dev), 54)
    _assert_iteration_agrees(res)


def test_composite_renders_depth_ties_long_lists_and_ragged_images(dev):
    """The merge rule under stress: scene and human Gaussians at EXACTLY the same depths (the scene wins ties, as its indices
    precede the human's in the concatenation), lists of several hundred entries per sub-tile (many 64-entry windows per
    merge, one source running out first), sub-tiles with only one source, a 75 x 100 image, huge splats over all cells."""
    H, W, f = 75, 100, 130.0
    g = torch.Generator().manual_seed(71)
    scene = scenes.dist_a_random(4000, H, W, seed=72, focal=f, z_range=(2.0, 5.0))
    human = scenes.dist_a_random(2500, H, W, seed=73, focal=f, z_range=(2.0, 5.0))
    # depth ties: the camera looks down +z from the origin, so equal z = equal depth bits
    zs = torch.linspace(2.2, 4.8, 40)
    scene['mean_3d'][:1200, 2] = zs[torch.randint(0, 40, (1200,), generator=g)]
    human['mean_3d'][:900, 2] = zs[torch.randint(0, 40, (900,), generator=g)]
    # deep lists: semi-transparent, fairly large
    scene['opacity'][:] = 0.02 + 0.1 * torch.rand(4000, 1, generator=g)
    human['opacity'][:] = 0.02 + 0.1 * torch.rand(2500, 1, generator=g)
    scene['scale'][:2000] *= 3.0
    human['scale'][:600] *= 4.0
    human['mean_3d'][600:, 0] = human['mean_3d'][600:, 0].abs() * 0.5 + 0.1        # the rest of the human on the right half only
    scene['scale'][-3:] = torch.tensor([0.6, 0.5, 0.4])                              # splats over every cell
    refined = {k: v.clone() for k, v in human.items()}
    refined['rgb'] = torch.rand(2500, 3, generator=g)
    res = _iteration_case(dev, scene, human, refined, H, W, f, torch.rand(3, generator=g).to(dev), 74)
    _assert_iteration_agrees(res)


def test_composite_copies_the_scene_where_the_human_is_absent(dev):
    """Where source B (the human) has no entry in a sub-tile, the composite's pixels are source A's own render (same list,
    same arithmetic, equal backgrounds): compose.hip empties such lists and render_fwd copies A's pixels
    (ExaRasterComposeJob.a_color).  A small avatar in front of a scene that fills the image: most sub-tiles take the copy.
    Bit-identical to blending everything again (knob off), to the concatenated render, in every image plane and gradient;
    with a DIFFERENT background on the scene render the device-side check must switch the copy off."""
    from exavatar_release_amd.renderer import _raster_job, _output_dict
    from exavatar_release_amd.rasterizer import rasterize_composites, rasterize_gaussians_batch
    H, W, f = 192, 256, 260.0
    scene = scenes.dist_a_random(6000, H, W, seed=91, focal=f, z_range=(3.0, 8.0))
    human = scenes.dist_b_avatar(3000, seed=92)
    human['mean_3d'] = human['mean_3d'] * 0.45 + torch.tensor([0.0, 0.0, 1.6])     # small, in the middle, in front of the scene
    human['scale'] = human['scale'] * 0.45
    cam = {k: t.to(dev) for k, t in scenes.neutral_camera(H, W, focal=f).items()}
    g = torch.Generator().manual_seed(93)
    G = [torch.randn(3, H, W, generator=g).to(dev) for _ in range(2)]
    Gd = torch.randn(1, H, W, generator=g).to(dev)

    def run(scene_bg, reuse):
        exa.config.compose_reuse_source = reuse
        try:
            s, h = _to(scene, dev), _to(human, dev)
            plain = [_raster_job(s, (H, W), cam, scene_bg), _raster_job(h, (H, W), cam, None)]
            outs, handles = rasterize_gaussians_batch(plain, keep_keys=True)
            comp = [_raster_job(h, (H, W), cam, None)]
            co = rasterize_composites([(handles[0], handles[1])], comp)[0]
            o = _output_dict(comp[0], co)
            ((o['img'] * G[0]).sum() + (o['depthmap'] * Gd).sum() + (outs[0][0] * G[1]).sum()).backward()
            torch.cuda.synchronize()
            return [o['img'].detach().clone(), o['depthmap'].detach().clone(), o['mask'].detach().clone(), o['radius'].clone()], \
                [h[k].grad.clone() for k in KEYS] + [s[k].grad.clone() for k in KEYS] + [o['mean_2d'].grad.clone()]
        finally:
            exa.config.compose_reuse_source = True
    white = None
    red = torch.tensor([1.0, 0.0, 0.0], device=dev)
    base_p, base_g = run(white, False)
    # how much of the image the human leaves alone: the copy must matter in this test
    mask = base_p[2][0]
    for name, (p_, g_) in (('copy', run(white, True)), ('other background: no copy', run(red, True))):
        for i, (x, y) in enumerate(zip(p_, base_p)):
            assert torch.equal(x, y), (name, 'plane', i)
        for i, (x, y) in enumerate(zip(g_[:5] + g_[10:], base_g[:5] + base_g[10:])):        # (the scene's own gradients depend on its bg)
            assert torch.equal(x, y), (name, 'grad', i)
    # ... and the reference's formulation: one render of the concatenation
    s, h = _to(scene, dev), _to(human, dev)
    cat = {k: torch.cat((s[k].detach(), h[k])) for k in KEYS}
    ref = exa.GaussianRenderer()(cat, (H, W), cam, None)
    assert torch.equal(ref['img'].detach(), base_p[0]) and torch.equal(ref['depthmap'].detach(), base_p[1])
    # the header of a composite with the copy names far fewer batch slots than the merged lists of the whole image would
    assert float((mask > 0).float().mean()) > 0.5       # (the scene covers most pixels; the human only a few sub-tiles)


def test_composite_render_agrees_with_the_oracle(dev):
    """The composite against the CPU oracle on the concatenation (not only against the HIP path's own concatenated render)."""
    H, W, f = 96, 128, 150.0
    scene = scenes.dist_a_random(1500, H, W, seed=81, focal=f)
    human = scenes.dist_a_random(800, H, W, seed=82, focal=f, z_range=(2.0, 4.0))
    cam = scenes.ring_camera(H, W, 2, 9, radius=3.0, center=(0.0, 0.0, 3.0), focal=f)
    camd = {k: t.to(dev) for k, t in cam.items()}
    g = torch.Generator().manual_seed(83)
    G = torch.randn(3, H, W, generator=g)
    s, h, r = _to(scene, dev), _to(human, dev), _to(human, dev)
    out = exa.render_iteration(exa.GaussianRenderer(), s, h, r, (H, W), camd, torch.ones(3, device=dev))
    o = out['scene_human']
    (o['img'] * G.to(dev)).sum().backward()
    h3 = {k: v.clone().requires_grad_(True) for k, v in human.items()}
    ref = ro.render({k: torch.cat((scene[k], h3[k])) for k in KEYS}, (H, W), cam, torch.ones(3), return_aux=True)
    (ref['img'] * G).sum().backward()
    amb = ro.ambiguous_pixel_mask(ref['aux'], H, W)
    assert_image_close(o['img'], ref['img'], amb, 'img')
    assert_image_close(o['depthmap'], ref['depthmap'], amb, 'depth')
    assert torch.equal(o['radius'].cpu(), ref['radius'])
    nS = scene['mean_3d'].shape[0]
    near = gaussians_near_pixels(ref['aux']['pre'], amb)[nS:]
    for k in KEYS:
        assert_grads_close(h[k].grad, h3[k].grad, k, near,
                           abs_scale=rotation_grad_scale(h3['scale'], h3['scale'].grad) if k == 'rotation' else 0.0)
    assert_grads_close(o['mean_2d'].grad, ref['mean_2d'].grad[nS:], 'mean_2d', near)
    assert all(s[k].grad is None for k in KEYS)          # only the composite was differentiated: the scene is a constant of it


def test_needle_conics_keep_the_power_guard(dev):
    """csrc/blend.h conic_safe: a conic with lambda_min / lambda_max below ~2.5e-6 (a needle of > 600 : 1 behind the 0.3 px^2
    low-pass floor) is the one kind of splat whose groups keep upstream's `power > 0 -> skip` compare, as a fix-up behind
    the evaluation; every other group runs without it.  Needles of 350 .. 3 000 px among ordinary splats, against the C
    oracle at the usual bar (measured: 1.3e-5)."""
    H, W, f = 40, 56, 100.0
    a = scenes.dist_a_random(60, H, W, seed=5, focal=f, z_range=(2.0, 4.0))
    g = torch.Generator().manual_seed(77)
    for i, s in enumerate((12.0, 30.0, 100.0, 40.0)):
        a['scale'][i] = torch.tensor([s, 1e-4, 1e-4])
        a['mean_3d'][i] = torch.tensor([0.1 * i - 0.2, 0.05 * i - 0.1, 3.0])
        q = torch.randn(4, generator=g)
        a['rotation'][i] = q / q.norm()
        a['opacity'][i] = 0.6
    cam, bg = scenes.neutral_camera(H, W, focal=f), torch.rand(3, generator=g)
    G = torch.randn(3, H, W, generator=g)
    ag = _to(a, dev)
    out = exa.GaussianRenderer()(ag, (H, W), {k: v.to(dev) for k, v in cam.items()}, bg.to(dev))
    (out['img'] * G.to(dev)).sum().backward()
    c = co.render(a, (H, W), cam, bg, dL_dimg=G)
    t = {k: v.clone().requires_grad_(True) for k, v in a.items()}
    r = ro.render(t, (H, W), cam, bg, return_aux=True)
    assert torch.equal(out['radius'].cpu(), c['radius'])
    assert int((c['radius'][:4] > 300).sum()) >= 3, 'the needles must be on screen as needles'
    amb = ro.ambiguous_pixel_mask(r['aux'], H, W) | (c['pixel_margin'] < 1e-4)
    d_hip = (out['img'].detach().cpu() - c['img']).abs().amax(0)
    d_cpu = (r['img'].detach() - c['img']).abs().amax(0)
    record_stats('needles', {'hip_vs_c': float(d_hip[~amb].max()), 'torch_vs_c': float(d_cpu[~amb].max()), 'n_ambiguous': int(amb.sum())})
    assert float(d_hip[~amb].max()) <= IMG_TOL, (float(d_hip[~amb].max()), float(d_cpu[~amb].max()))
    for k in KEYS:
        assert bool(torch.isfinite(ag[k].grad).all()), k
