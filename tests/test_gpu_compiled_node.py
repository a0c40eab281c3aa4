"""The compiled autograd node (``csrc/torch_binding.cpp`` -> ``_exa_torch.so``) against the Python node it stands in for
(``rasterizer._Rasterize``): the same C-ABI calls on the same arena layouts, so every output and every gradient must be equal
BIT FOR BIT -- precomputed colours and in-kernel SH, depth / alpha gradients, fused densification statistics, non-leaf inputs,
an overflow repaired inside the call, ``no_grad`` renders, exact (two-stage) and capacity mode -- and the calls it does not cover
(non-contiguous inputs, settings that live on the host) must fall through to the Python node unnoticed."""
import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import rasterizer as rz
from exavatar_release_amd import scenes
from exavatar_release_amd.camera import make_raster_matrices

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    from exavatar_release_amd import _lib
    _lib.load()
    exa.config.compiled_node = 'require'
    assert rz._compiled_node(), 'the compiled autograd node must be built on a GPU box (python -m exavatar_release_amd.build)'
    exa.config.compiled_node = 'auto'
    return torch.device('cuda:0')


def _settings(cam, H, W, bg, dev, sh_degree=0):
    tanx, tany, view, proj, campos = make_raster_matrices(cam, (H, W))
    return exa.GaussianRasterizationSettings(H, W, tanx, tany, bg, 1.0, view.to(dev).contiguous(), proj.to(dev).contiguous(),
                                             sh_degree, campos.to(dev).contiguous(), False, False)


def _run(a, st, G, Gd, Ga, sh=None, dens=None, through_ops=False):
    """One render fwd + bwd through ``GaussianRasterizer`` on fresh leaves; returns (outputs, gradients, node that ran)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in a.items()}
    m2 = torch.zeros_like(leaves['mean_3d'], requires_grad=True)
    shl = sh.detach().clone().requires_grad_(True) if sh is not None else None
    m3 = leaves['mean_3d'] * 1.0 + 0.0 if through_ops else leaves['mean_3d']        # a non-leaf input (LBS in front of the render)
    n0 = rz.compiled_calls
    if dens is None:
        color, radii, depth, alpha = exa.GaussianRasterizer(st)(
            means3D=m3, means2D=m2, opacities=leaves['opacity'], shs=shl,
            colors_precomp=None if sh is not None else leaves['rgb'], scales=leaves['scale'], rotations=leaves['rotation'])
    else:
        color, radii, depth, alpha = rz.rasterize_gaussians(m3, m2, shl, None if sh is not None else leaves['rgb'], leaves['opacity'],
                                                            leaves['scale'], leaves['rotation'], None, st, densify_stats=dens)
    is_vis = rz.take_is_vis()[0]
    loss = (color * G).sum()
    if Gd is not None:
        loss = loss + (depth * Gd).sum() + (alpha * Ga).sum()
    loss.backward()
    g = {'means3D': leaves['mean_3d'].grad, 'means2D': m2.grad, 'opacities': leaves['opacity'].grad, 'scales': leaves['scale'].grad,
         'rotations': leaves['rotation'].grad}
    g['colour'] = shl.grad if sh is not None else leaves['rgb'].grad
    return (color.detach(), radii, depth.detach(), alpha.detach(), is_vis), g, rz.compiled_calls - n0


def _same(x, y):
    for a, b in zip(x[0], y[0]):
        assert torch.equal(a, b)
    for k in x[1]:
        assert torch.equal(x[1][k], y[1][k]), k


@pytest.mark.parametrize('use_sh', [False, True])
@pytest.mark.parametrize('image_grads', ['colour', 'all'])
@pytest.mark.parametrize('mode', ['auto', 'exact'])
def test_compiled_node_equals_the_python_node_bit_for_bit(dev, use_sh, image_grads, mode):
    H, W, f, P = 144, 176, 240.0, 5000 + 7
    a = {k: v.to(dev) for k, v in scenes.dist_b_avatar(P, seed=5).items()}
    sh = scenes.sh_from_rgb(a['rgb'].cpu(), 2, seed=3, rest_sigma=0.3).to(dev).contiguous() if use_sh else None
    g = torch.Generator().manual_seed(6)
    G, Gd, Ga = (torch.randn(n, H, W, generator=g).to(dev) for n in (3, 1, 1))
    if image_grads == 'colour':
        Gd = Ga = None
    bg = torch.rand(3, generator=g).to(dev)
    exa.config.mode = mode
    for v in (0, 7, 19):
        st = _settings(scenes.ring_camera(H, W, v, 24, focal=f), H, W, bg, dev, 2 if use_sh else 0)
        exa.config.compiled_node = 'off'
        ref = _run(a, st, G, Gd, Ga, sh)            # (auto: also measures the capacity of this shape, the first call is exact)
        ref = _run(a, st, G, Gd, Ga, sh)
        assert ref[2] == 0
        exa.config.compiled_node = 'auto'
        got = _run(a, st, G, Gd, Ga, sh)
        assert got[2] == 1, 'the compiled node did not take a call it covers'
        _same(ref, got)
        got = _run(a, st, G, Gd, Ga, sh, through_ops=True)
        assert got[2] == 1
        _same(ref, got)


def test_compiled_node_updates_the_fused_densification_statistics(dev):
    H, W, P = 128, 160, 3000
    a = {k: v.to(dev) for k, v in scenes.dist_a_random(P, H, W, seed=11, focal=200.0).items()}
    G = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2)).to(dev)
    st = _settings(scenes.neutral_camera(H, W, focal=200.0), H, W, torch.ones(3, device=dev), dev)
    stats = {}
    for how in ('off', 'auto'):
        exa.config.compiled_node = how
        dens = tuple(torch.full((P,), float(i), device=dev) for i in range(3))
        for _ in range(3):
            out = _run(a, st, G, None, None, dens=dens)
        assert out[2] == (1 if how == 'auto' else 0)
        stats[how] = (out, dens)
    _same(stats['off'][0], stats['auto'][0])
    for x, y in zip(stats['off'][1], stats['auto'][1]):
        assert torch.equal(x, y)
    assert float(stats['auto'][1][1].max()) == 1.0 + 3.0          # track_cnt: three backwards of visible Gaussians


def test_compiled_node_repairs_an_overflow_inside_the_call(dev):
    H, W, P = 144, 176, 6000
    a = {k: v.to(dev) for k, v in scenes.dist_b_avatar(P, seed=9).items()}
    G = torch.randn(3, H, W, generator=torch.Generator().manual_seed(4)).to(dev)
    st = _settings(scenes.ring_camera(H, W, 3, 24, focal=240.0), H, W, torch.ones(3, device=dev), dev)
    exa.config.compiled_node = 'off'
    exa.config.mode = 'exact'
    ref = _run(a, st, G, None, None)
    exa.config.mode = 'capacity'
    key = (dev.index or 0, P, H, W)
    need = rz._seen_D[key]
    exa.config.compiled_node = 'auto'
    exa.config.min_capacity = 64
    for cap in (64, 1024, need - 64):
        exa.config.fixed_capacity = cap
        n0 = len(rz.overflow_events)
        got = _run(a, st, G, None, None)
        assert got[2] == 1
        assert len(rz.overflow_events) == n0 + 1 and rz.overflow_events[-1][1:] == (need, (cap + 63) // 64 * 64, 'retried')
        _same(ref, got)
    exa.config.fixed_capacity = need
    n0 = len(rz.overflow_events)
    _same(ref, _run(a, st, G, None, None))
    assert len(rz.overflow_events) == n0


def test_calls_the_compiled_node_does_not_cover_fall_through(dev):
    H, W, P = 96, 128, 2000
    a = {k: v.to(dev) for k, v in scenes.dist_a_random(P, H, W, seed=3, focal=150.0).items()}
    G = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    st = _settings(scenes.neutral_camera(H, W, focal=150.0), H, W, torch.ones(3, device=dev), dev)
    key = (dev.index or 0, P, H, W)
    rz._seen_D.pop(key, None)
    first = _run(a, st, G, None, None)
    assert first[2] == 1 and key in rz._seen_D            # first call of a shape: measured in exact mode (two stages), by either node
    second = _run(a, st, G, None, None)
    assert second[2] == 1
    _same(first, second)
    # a no_grad render: 'auto' sizes it exactly, 'capacity' from the memo; no context is kept either way
    with torch.no_grad():
        outs = {}
        for how in ('off', 'auto'):
            exa.config.compiled_node = how
            for mode in ('auto', 'capacity'):
                exa.config.mode = mode
                n0 = rz.compiled_calls
                outs[how, mode] = exa.GaussianRasterizer(st)(means3D=a['mean_3d'], means2D=torch.zeros_like(a['mean_3d']),
                                                             opacities=a['opacity'], colors_precomp=a['rgb'], scales=a['scale'],
                                                             rotations=a['rotation'])
                assert rz.compiled_calls == n0 + (how == 'auto')
        exa.config.mode = 'auto'
    for o in outs.values():
        for x, y in zip(o, outs['off', 'auto']):
            assert torch.equal(x, y)
        assert torch.equal(o[0], first[0][0]) and not o[0].requires_grad
    # non-contiguous input: converted by the Python node
    b = dict(a)
    b['scale'] = a['scale'].t().contiguous().t()
    assert not b['scale'].is_contiguous()
    third = _run(b, st, G, None, None)
    assert third[2] == 0
    _same(first, third)
    # settings whose camera lives on the host: converted by the Python node
    st_cpu = st._replace(bg=torch.ones(3))
    assert _run(a, st_cpu, G, None, None)[2] == 0
    # upstream's error messages come from the Python surface either way
    with pytest.raises(Exception, match='excatly one of either SHs or precomputed colors'):
        exa.GaussianRasterizer(st)(means3D=a['mean_3d'], means2D=torch.zeros_like(a['mean_3d']), opacities=a['opacity'],
                                   scales=a['scale'], rotations=a['rotation'])


def test_five_live_contexts_and_retain_graph(dev):
    """``avatar/main/model.py:130-162`` keeps five forward contexts alive before one backward runs through all of them."""
    H, W, P = 128, 128, 4000
    a = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_b_avatar(P, seed=1).items()}
    G = torch.randn(3, H, W, generator=torch.Generator().manual_seed(8)).to(dev)
    sts = [_settings(scenes.ring_camera(H, W, v, 10, focal=200.0), H, W, torch.ones(3, device=dev), dev) for v in range(5)]

    def loss():
        outs = [exa.GaussianRasterizer(st)(means3D=a['mean_3d'], means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
                                           opacities=a['opacity'], colors_precomp=a['rgb'], scales=a['scale'], rotations=a['rotation'])
                for st in sts]
        return sum((o[0] * G).sum() * (i + 1) for i, o in enumerate(outs))
    grads = {}
    for how in ('off', 'auto', 'auto'):
        exa.config.compiled_node = how
        n0 = rz.compiled_calls
        L = loss()
        g1 = torch.autograd.grad(L, list(a.values()), retain_graph=True)
        g2 = torch.autograd.grad(L, list(a.values()))
        for x, y in zip(g1, g2):
            assert torch.equal(x, y)
        grads[how] = g1
        if how == 'auto':
            assert rz.compiled_calls - n0 in (0, 5)
    assert rz.compiled_calls - n0 == 5
    for x, y in zip(grads['off'], grads['auto']):
        assert torch.equal(x, y)


def test_training_loop_through_the_compiled_node_equals_the_python_node(dev):
    """A loop shaped like ``avatar/main/train.py:41-57`` with the reference's own structure -- FIVE single renders per iteration
    (``avatar/main/model.py:130-162``: scene, human, cat(scene.detach(), human), refined, cat(scene.detach(), refined)), one
    backward through all five live contexts, Adam, the Gaussian count of the scene changing every 25 iterations, the capacity memo
    forgotten now and then (overflows repaired inside the call), a ``no_grad`` evaluation render every 10th iteration -- once
    through the compiled node, once through the Python node: the same parameters bit for bit."""
    H, W, f = 96, 128, 170.0
    keys = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')
    scene0 = scenes.dist_a_random(1200, H, W, seed=31, focal=f)
    human0 = scenes.dist_a_random(900, H, W, seed=32, focal=f, z_range=(2.0, 4.0))
    cams = [{k: v.to(dev) for k, v in scenes.ring_camera(H, W, 5 * v, 40, radius=3.2, center=(0.0, 0.0, 3.0), focal=f).items()} for v in range(4)]
    bg = torch.tensor([0.2, 0.5, 0.3], device=dev)
    g = torch.Generator().manual_seed(9)
    Gs = [torch.randn(3, H, W, generator=g).to(dev) for _ in range(5)]
    rend = exa.GaussianRenderer()

    def run(how):
        exa.config.compiled_node = how
        exa.config.mode, exa.config.capacity_growth, exa.config.min_capacity = 'auto', 1.05, 64
        rz._seen_D.clear()
        n_ev = len(rz.overflow_events)
        n0 = rz.compiled_calls
        sets = [{k: torch.nn.Parameter(d[k].to(dev).clone()) for k in keys} for d in (scene0, human0, human0)]
        make_opt = lambda: torch.optim.Adam([p for s_ in sets for p in s_.values()], lr=1e-3, eps=1e-15)      # noqa: E731
        opt = make_opt()
        losses, evals = [], []
        for i in range(80):
            cam = cams[(i * 3) % len(cams)]
            if i % 13 == 12:
                for key in list(rz._seen_D):
                    rz._seen_D[key] = int(rz._seen_D[key] * 0.6)
            scene, human, refined = sets
            cat = lambda a_, b_: {k: torch.cat((a_[k].detach(), b_[k])) for k in keys}      # noqa: E731
            outs = [rend(scene, (H, W), cam), rend(human, (H, W), cam, bg), rend(cat(scene, human), (H, W), cam),
                    rend(refined, (H, W), cam, bg), rend(cat(scene, refined), (H, W), cam)]
            loss = sum((o['img'] * G).sum() for o, G in zip(outs, Gs)) * 1e-3
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            with torch.no_grad():
                for s_ in sets:
                    s_['opacity'].clamp_(0.01, 0.99); s_['scale'].clamp_(1e-4, 1.0); s_['rgb'].clamp_(0.0, 1.0)
            losses.append(float(loss.detach()))
            if i % 10 == 9:
                with torch.no_grad():
                    evals.append(float(rend(human, (H, W), cam, bg)['img'].mean()))
            if i % 25 == 24:                       # the scene grows / shrinks: new tensors, new P, a new optimizer
                P = scene['mean_3d'].shape[0]
                keep = torch.arange(P, device=dev) % 7 != (i // 25)
                top = torch.arange(0, P, 5, device=dev)
                sets[0] = {k: torch.nn.Parameter(torch.cat((v.detach()[keep], v.detach()[top] + (0.01 if k == 'mean_3d' else 0.0))).contiguous())
                           for k, v in scene.items()}
                opt = make_opt()
        torch.cuda.synchronize()
        return ([p.detach().clone() for s_ in sets for p in s_.values()], losses, evals, rz.compiled_calls - n0,
                [e[3] for e in rz.overflow_events[n_ev:]] if len(rz.overflow_events) >= n_ev else [])
    ref = run('off')
    got = run('auto')
    assert ref[3] == 0 and got[3] >= 80 * 5
    assert got[1] == ref[1] and got[2] == ref[2]
    for x, y in zip(got[0], ref[0]):
        assert x.shape == y.shape and torch.equal(x, y)
    assert 'retried' in got[4] and set(got[4]) == {'retried'}
