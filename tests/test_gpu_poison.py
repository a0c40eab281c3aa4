"""The library promises that every workspace section is written by its producer (or zero-filled by the binning) before
anybody reads it -- the workspaces come from ``torch.empty``.  ``config.poison`` fills them with 0xFF (NaN as floats, huge
as indices) before the kernels see them; here: capacity-mode renders WITH SLACK in the instance buffer (the regions behind the
instances in use are never written), single, as a five-render iteration with composites, and replayed from hipGraphs, must
equal exact-mode renders bit for bit.  (The whole suite also runs under it: ``EXA_TEST_POISON=1``, tests/conftest.py.)"""
import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import rasterizer as rz, scenes

pytestmark = pytest.mark.gpu

KEYS = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')
H, W, F = 128, 160, 170.0


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    from exavatar_release_amd import _lib
    _lib.load()
    exa.config.mode, exa.config.fixed_capacity = 'exact', None
    return torch.device('cuda:0')


def _leaves(d, dev):
    return {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in d.items()}


def _assert_same_grads(a, b, what):
    for k in KEYS:
        assert not bool(torch.isnan(a[k].grad).any()), (what, k, 'NaN: a poisoned section was read')
        assert torch.equal(a[k].grad, b[k].grad), (what, k)


def test_poisoned_workspaces_do_not_show_through(dev):
    scene = scenes.dist_a_random(3000, H, W, seed=51, focal=F)
    human = scenes.dist_a_random(1500, H, W, seed=52, focal=F, z_range=(2.0, 4.0))
    cam = {k: v.to(dev) for k, v in scenes.neutral_camera(H, W, focal=F).items()}
    bg = torch.rand(3, device=dev)
    G = torch.randn(3, H, W, device=dev)
    rend = exa.GaussianRenderer()
    ref = _leaves(scene, dev)
    ref_out = rend(ref, (H, W), cam, bg)
    (ref_out['img'] * G).sum().backward()
    r3 = [_leaves(scene, dev), _leaves(human, dev), _leaves(human, dev)]
    out3 = exa.render_iteration(rend, *r3, (H, W), cam, bg)
    sum((out3[k]['img'] * G).sum() for k in exa.ITERATION_RENDERS).backward()
    ns, nh = rz._seen_D[(dev.index or 0, 3000, H, W)], rz._seen_D[(dev.index or 0, 1500, H, W)]
    exa.config.poison = True
    try:
        for slack in (1.0, 1.5, 3.0):
            exa.config.mode, exa.config.fixed_capacity = 'capacity', int(ns * slack)
            a = _leaves(scene, dev)
            out = rend(a, (H, W), cam, bg)
            (out['img'] * G).sum().backward()
            assert torch.equal(out['img'].detach(), ref_out['img'].detach()) and torch.equal(out['radius'], ref_out['radius'])
            _assert_same_grads(a, ref, 'single render, slack %.1f' % slack)
        for slack in (1.0, 2.0):
            exa.config.mode, exa.config.fixed_capacity = 'capacity', [int(ns * slack), int(nh * slack), int(nh * slack)]
            s = [_leaves(scene, dev), _leaves(human, dev), _leaves(human, dev)]
            out = exa.render_iteration(rend, *s, (H, W), cam, bg)
            sum((out[k]['img'] * G).sum() for k in exa.ITERATION_RENDERS).backward()
            for k in exa.ITERATION_RENDERS:
                assert torch.equal(out[k]['img'].detach(), out3[k]['img'].detach()), k
            for i, name in enumerate(('scene', 'human', 'refined human')):
                _assert_same_grads(s[i], r3[i], 'iteration, slack %.1f, %s' % (slack, name))
        # StaticRender (the C ABI with static storage, bench.py's headline path): ONE set of workspaces reused by views with very
        # different instance counts, allocated through the same poisoned allocator -- stale sections would show here
        from exavatar_release_amd.renderer import _raster_job
        cam_b = {k: v.to(dev) for k, v in scenes.ring_camera(H, W, 3, 8, radius=2.0, center=(0.0, 0.0, 4.0), focal=F).items()}
        exa.config.mode, exa.config.fixed_capacity = 'exact', None
        exa.config.poison = False
        ref_b = _leaves(scene, dev)
        ref_b_out = rend(ref_b, (H, W), cam_b, bg)
        (ref_b_out['img'] * G).sum().backward()
        exa.config.poison = True
        flat = {k: v.detach().to(dev).contiguous() for k, v in scene.items()}
        sts = [_raster_job(flat, (H, W), c, bg)['raster_settings'] for c in (cam, cam_b)]
        with exa.StaticRender(flat['mean_3d'], flat['opacity'], flat['scale'], flat['rotation'], colors_precomp=flat['rgb'],
                              image_size=(H, W), capacity=int(ns * 3.0), slots=2) as sr:
            vs = [sr.add_view(st, dL_dcolor=G) for st in sts]
            assert [sr.add_grad_outputs() for _ in range(2)] == [0, 1]
            for order in ((0, 1), (1, 0), (0, 0), (1, 1)):
                for slot, v in enumerate(order):
                    sr.forward(vs[v], slot=slot)
                    sr.backward(slot, slot=slot)
                sr.check()
                for slot, v in enumerate(order):
                    want_out, want = (ref_out, ref) if v == 0 else (ref_b_out, ref_b)
                    assert torch.equal(sr.outputs(slot)['color'], want_out['img'].detach()), ('static', order, slot)
                    g = sr.grad_outputs(slot)
                    for k, n in (('mean_3d', 'means3D'), ('scale', 'scales'), ('rotation', 'rotations'), ('opacity', 'opacities'),
                                 ('rgb', 'colors_precomp')):
                        assert not bool(torch.isnan(g[n]).any()), ('static', k, 'NaN: a poisoned section was read')
                        assert torch.equal(g[n].view_as(want[k].grad), want[k].grad), ('static', order, slot, k)
        exa.config.mode, exa.config.fixed_capacity = 'exact', None
        with exa.GraphedIteration((H, W), dev) as it:
            for rep in range(2):
                s = [_leaves(scene, dev), _leaves(human, dev), _leaves(human, dev)]
                out = it(*s, cam, bg)
                imgs = [out[k]['img'].detach().clone() for k in exa.ITERATION_RENDERS]
                sum((out[k]['img'] * G).sum() for k in exa.ITERATION_RENDERS).backward()
                torch.cuda.synchronize()
                for k, im in zip(exa.ITERATION_RENDERS, imgs):
                    assert torch.equal(im, out3[k]['img'].detach()), ('graphed', k)
                for i, name in enumerate(('scene', 'human', 'refined human')):
                    _assert_same_grads(s[i], r3[i], 'graphed iteration, replay %d, %s' % (rep, name))
    finally:
        exa.config.poison = False
        exa.config.mode, exa.config.fixed_capacity = 'exact', None
