"""GPU tests of two things the iteration path does without kernels of their own (round 4):

* ``is_vis`` of every output dict is written by the per-Gaussian forward kernel next to ``radii``
  (``ExaRasterForwardJob.is_vis``) -- it must be ``radius > 0`` (reference ``avatar/common/nets/module.py:645``) for single
  renders, batches, constant-prefix renders and the five renders of an iteration; the composites' ``radius`` / ``is_vis`` are
  concatenations built on first access and must be what the reference's concatenated render returns.
* the gradients a composite render produces for the human's tensors are added INSIDE the per-Gaussian kernel of the human's
  own render (``ExaRasterBackwardJob.accumulate``, ``config.fold_composite_grads``) instead of by autograd: bit-identical to
  autograd's sum for every pattern of outputs the loss reads, with SH inputs, for a second backward over a retained graph.

/root/reference is never read here."""
import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import scenes

pytestmark = pytest.mark.gpu

KEYS = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')
H, W, F = 128, 160, 170.0


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    from exavatar_release_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _config():
    saved = (exa.config.mode, exa.config.fixed_capacity, exa.config.fold_composite_grads)
    exa.config.mode, exa.config.fixed_capacity = 'exact', None
    yield
    exa.config.mode, exa.config.fixed_capacity, exa.config.fold_composite_grads = saved


def _sets(n_scene, n_human, seed, dev, sh=False):
    scene = scenes.dist_a_random(n_scene, H, W, seed=seed, focal=F)
    human = scenes.dist_a_random(n_human, H, W, seed=seed + 1, focal=F, z_range=(2.0, 4.0))
    g = torch.Generator().manual_seed(seed + 2)
    refined = {k: (v + 0.01 * torch.randn(v.shape, generator=g) if k == 'mean_3d' else v.clone()) for k, v in human.items()}
    sets = [{k: v.to(dev) for k, v in d.items()} for d in (scene, human, refined)]
    if sh:
        for d in sets:
            rgb = d.pop('rgb')
            d['sh'] = torch.cat(((rgb[:, None] - 0.5) / 0.28209479, 0.2 * torch.randn(rgb.shape[0], 8, 3, generator=g).to(dev)), 1)
            d['sh_degree'] = 2
    return sets


def _leaves(sets):
    return [{k: (v.detach().clone().requires_grad_(True) if torch.is_tensor(v) else v) for k, v in d.items()} for d in sets]


def _cam(i, dev):
    return {k: t.to(dev) for k, t in scenes.ring_camera(H, W, i, 40, radius=3.2, center=(0.0, 0.0, 3.0), focal=F).items()}


def _behind(cam, n):
    """n world positions two units BEHIND the camera (x_cam = R x + t, reference module.py:594-611)."""
    R, t = cam['R'].reshape(3, 3), cam['t'].reshape(3)
    x_cam = torch.tensor([0.0, 0.0, -2.0], device=R.device).expand(n, 3) + 0.1 * torch.randn(n, 3, device=R.device)
    return (x_cam - t) @ R                      # = R^T (x_cam - t), row-wise


def test_is_vis_is_written_by_the_forward_kernel(dev):
    rend = exa.GaussianRenderer()
    s, h, r = _sets(900, 700, 5, dev)
    cam, bg = _cam(3, dev), torch.rand(3, device=dev)
    # some Gaussians behind the camera: is_vis must be False there
    s['mean_3d'][:50] = _behind(cam, 50)
    h['mean_3d'][:40] = _behind(cam, 40)
    out = rend(s, (H, W), cam, bg)
    assert out['is_vis'].dtype == torch.bool and out['is_vis'].shape == out['radius'].shape
    assert torch.equal(out['is_vis'], out['radius'] > 0)
    assert not bool(out['is_vis'][:50].any()) and bool(out['is_vis'].any())
    assert list(out.keys()) == ['img', 'depthmap', 'mask', 'mean_2d', 'is_vis', 'radius']
    with torch.no_grad():
        out = rend(h, (H, W), cam, bg)
    assert torch.equal(out['is_vis'], out['radius'] > 0) and not bool(out['is_vis'][:40].any())
    # a batch, one job with a constant prefix (radius covers cat(prefix, own))
    many = exa.render_many(rend, [(s, (H, W), cam), (h, (H, W), cam, bg), (h, (H, W), cam, None, None, s)])
    for o, n in zip(many, (900, 700, 1600)):
        assert o['is_vis'].dtype == torch.bool and o['is_vis'].shape == (n,)
        assert torch.equal(o['is_vis'], o['radius'] > 0)
    assert torch.equal(many[2]['is_vis'], torch.cat((many[0]['is_vis'], many[1]['is_vis'])))


@pytest.mark.parametrize('merge', [True, False])
def test_iteration_is_vis_and_composite_entries(dev, merge):
    rend = exa.GaussianRenderer()
    s, h, r = _sets(800, 600, 11, dev)
    cam, bg = _cam(5, dev), torch.rand(3, device=dev)
    s['mean_3d'][:30] = _behind(cam, 30)
    out = exa.render_iteration(rend, s, h, r, (H, W), cam, bg, merge=merge)
    for name in exa.ITERATION_RENDERS:
        o = out[name]
        assert type(o) is dict and list(o.keys()) == ['img', 'depthmap', 'mask', 'mean_2d', 'is_vis', 'radius'], name
        assert torch.equal(o['is_vis'], o['radius'] > 0), name
        assert o['is_vis'].dtype == torch.bool
    for comp, b in (('scene_human', 'human'), ('scene_human_refined', 'human_refined')):
        assert torch.equal(out[comp]['radius'], torch.cat((out['scene']['radius'], out[b]['radius'])))
        assert out[comp]['radius'].shape == (1400,)
        assert all(torch.is_tensor(v) for v in out[comp].values())


def _iteration(sets, cam, bg, G, which, dev, fold, second=False):
    exa.config.fold_composite_grads = fold
    rend = exa.GaussianRenderer()
    s, h, r = _leaves(sets)
    dens = tuple(torch.zeros(s['mean_3d'].shape[0], device=dev) for _ in range(3))
    out = exa.render_iteration(rend, s, h, r, (H, W), cam, bg, dens)
    loss = sum((out[n]['img'] * G[i]).sum() * (0.5 + 0.25 * i) for i, n in enumerate(exa.ITERATION_RENDERS) if n in which)
    if 'planes' in which:
        loss = loss + (out['scene_human']['mask'] * G[5][:1]).sum() + (out['human']['depthmap'] * G[5][1:2]).sum()
    loss.backward(retain_graph=second)
    if second:
        for t in (s, h, r):
            for k, v in t.items():
                if torch.is_tensor(v):
                    v.grad = None
        for n in exa.ITERATION_RENDERS:
            out[n]['mean_2d'].grad = None
        loss.backward()
    torch.cuda.synchronize()
    grads = {}
    for name, t in zip(('scene', 'human', 'refined'), (s, h, r)):
        for k, v in t.items():
            if torch.is_tensor(v):
                grads[name + '.' + k] = None if v.grad is None else v.grad.clone()
    for n in exa.ITERATION_RENDERS:
        g = out[n]['mean_2d'].grad
        grads[n + '.mean_2d'] = None if g is None else g.clone()
    return grads, [d.clone() for d in dens]


ALL = exa.ITERATION_RENDERS


@pytest.mark.parametrize('which', [ALL, ALL + ('planes',), ('scene_human', 'scene_human_refined'), ('scene', 'human', 'human_refined'),
                                   ('human', 'scene_human'), ('scene_human',), ('scene',)])
def test_folded_composite_gradients_are_autograds_sum_bit_for_bit(dev, which):
    sets = _sets(1200, 900, 21, dev)
    cam, bg = _cam(7, dev), torch.rand(3, device=dev)
    G = torch.randn(6, 3, H, W, device=dev)
    a, da = _iteration(sets, cam, bg, G, which, dev, fold=True)
    b, db = _iteration(sets, cam, bg, G, which, dev, fold=False)
    assert a.keys() == b.keys()
    for k in a:
        assert (a[k] is None) == (b[k] is None), k
        if a[k] is not None:
            assert torch.equal(a[k], b[k]), k
    assert all(torch.equal(x, y) for x, y in zip(da, db))
    # the gradients are real ones: a human that the loss sees got something
    if any(n != 'scene' for n in which if n != 'planes'):
        assert any(a[k] is not None and float(a[k].abs().sum()) > 0 for k in a if k.startswith(('human.', 'refined.')))


def test_folded_gradients_with_sh_inputs_and_a_second_backward(dev):
    sets = _sets(700, 500, 31, dev, sh=True)
    cam, bg = _cam(2, dev), torch.rand(3, device=dev)
    G = torch.randn(6, 3, H, W, device=dev)
    for second in (False, True):
        a, _ = _iteration(sets, cam, bg, G, ALL, dev, fold=True, second=second)
        b, _ = _iteration(sets, cam, bg, G, ALL, dev, fold=False, second=second)
        for k in a:
            assert (a[k] is None) == (b[k] is None), k
            if a[k] is not None:
                assert torch.equal(a[k], b[k]), (k, second)
        assert float(a['human.sh'].abs().sum()) > 0


def test_fold_leaves_nothing_behind_and_handles_foreign_tokens(dev):
    """The stash is consumed by the source's backward; a composite given no token (or sources of another batch) returns its
    gradients itself."""
    from exavatar_release_amd import renderer as rr
    from exavatar_release_amd.rasterizer import rasterize_composites, rasterize_gaussians_batch
    sets = _sets(600, 400, 41, dev)
    s, h, _r = _leaves(sets)
    cam, bg = _cam(4, dev), torch.rand(3, device=dev)
    G = torch.randn(2, 3, H, W, device=dev)

    def once(token_of):
        for t in (s, h):
            for v in t.values():
                v.grad = None
        plain = [rr._raster_job(s, (H, W), cam, None), rr._raster_job(h, (H, W), cam, bg)]
        outs, handles = rasterize_gaussians_batch(plain, keep_keys=True)
        comp = [rr._raster_job(h, (H, W), cam, None)]
        tok = {'own': handles.token, 'none': None}[token_of]
        co = rasterize_composites([(handles[0], handles[1])], comp, token=tok)[0]
        ((outs[1][0] * G[0]).sum() + (co[0] * G[1]).sum()).backward()
        torch.cuda.synchronize()
        assert all(j.stash is None for j in handles)
        return [h[k].grad.clone() for k in KEYS]

    a, b = once('own'), once('none')
    assert all(torch.equal(x, y) for x, y in zip(a, b))
