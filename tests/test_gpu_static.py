"""``StaticRender`` (exavatar_release_amd/static.py): the C ABI driven with static storage and pre-marshalled jobs must give
what the autograd surface gives -- same kernels, same bits -- for several cameras in turn, with precomputed colours and with
in-kernel SH, for training and no_grad renders, into caller-owned gradient arrays; and an overflowed render must raise."""
import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.camera import make_raster_matrices

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    from exavatar_release_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _settings(cam, H, W, bg, dev, sh_degree=0):
    tanx, tany, view, proj, campos = make_raster_matrices(cam, (H, W))
    return exa.GaussianRasterizationSettings(H, W, tanx, tany, bg, 1.0, view.to(dev).contiguous(), proj.to(dev).contiguous(),
                                             sh_degree, campos.to(dev).contiguous(), False, False)


def _reference(a, st, G, Gd, Ga, sh=None):
    """The autograd surface on fresh leaves: images, radii and gradients."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in a.items()}
    m2 = torch.zeros_like(leaves['mean_3d'], requires_grad=True)
    shl = sh.detach().clone().requires_grad_(True) if sh is not None else None
    color, radii, depth, alpha = exa.GaussianRasterizer(st)(
        means3D=leaves['mean_3d'], means2D=m2, opacities=leaves['opacity'], shs=shl,
        colors_precomp=None if sh is not None else leaves['rgb'], scales=leaves['scale'], rotations=leaves['rotation'])
    ((color * G).sum() + (depth * Gd).sum() + (alpha * Ga).sum()).backward()
    g = {'means3D': leaves['mean_3d'].grad, 'means2D': m2.grad, 'opacities': leaves['opacity'].grad, 'scales': leaves['scale'].grad,
         'rotations': leaves['rotation'].grad}
    if sh is not None:
        g['shs'] = shl.grad
    else:
        g['colors_precomp'] = leaves['rgb'].grad
    return color.detach(), depth.detach(), alpha.detach(), radii, g


@pytest.mark.parametrize('use_sh', [False, True])
def test_static_render_equals_the_autograd_surface_bit_for_bit(dev, use_sh):
    H, W, f, P = 144, 176, 240.0, 5000 + 7
    a = {k: v.to(dev) for k, v in scenes.dist_b_avatar(P, seed=5).items()}
    sh = scenes.sh_from_rgb(a['rgb'].cpu(), 2, seed=3, rest_sigma=0.3).to(dev).contiguous() if use_sh else None
    g = torch.Generator().manual_seed(6)
    G, Gd, Ga = (torch.randn(n, H, W, generator=g).to(dev) for n in (3, 1, 1))
    bg = torch.rand(3, generator=g).to(dev)
    sts = [_settings(scenes.ring_camera(H, W, v, 24, focal=f), H, W, bg, dev, 2 if use_sh else 0) for v in (0, 7, 19)]
    exa.config.mode = 'exact'
    refs = [_reference(a, st, G, Gd, Ga, sh) for st in sts]
    need = exa.required_capacity(a['mean_3d'], a['opacity'], a['scale'], a['rotation'],
                                 colors_precomp=None if use_sh else a['rgb'], shs=sh, settings=sts)
    with exa.StaticRender(a['mean_3d'], a['opacity'], a['scale'], a['rotation'], colors_precomp=None if use_sh else a['rgb'], shs=sh,
                          image_size=(H, W), capacity=need) as sr:
        views = [sr.add_view(st, dL_dcolor=G, dL_ddepth=Gd, dL_dalpha=Ga) for st in sts]
        mine = sr.add_grad_outputs()                                  # set 0: allocated by the object
        flat = torch.zeros(P * 3, device=dev)                         # set 1: dL/dmeans3D inside a caller-owned flat buffer
        other = sr.add_grad_outputs(means3D=flat.view(P, 3))
        for rounds in range(2):                                       # every camera twice: nothing left over from the call before
            for v, (color, depth, alpha, radii, gr) in zip(views, refs):
                out = other if rounds else mine
                sr.forward(v)
                sr.backward(out)
                sr.check()
                assert torch.equal(sr.color, color) and torch.equal(sr.depth, depth) and torch.equal(sr.alpha, alpha)
                assert torch.equal(sr.radii, radii)
                assert torch.equal(sr.is_vis, radii > 0)
                got = sr.grad_outputs(out)
                for k, ref in gr.items():
                    assert torch.equal(got[k].view_as(ref), ref), k
        assert torch.equal(flat.view(P, 3), refs[-1][4]['means3D'])


def test_static_no_grad_render_and_overflow_raises(dev):
    H, W, f, P = 128, 128, 200.0, 4000
    a = {k: v.to(dev) for k, v in scenes.dist_b_avatar(P, seed=8).items()}
    bg = torch.ones(3, device=dev)
    st = _settings(scenes.ring_camera(H, W, 3, 24, focal=f), H, W, bg, dev)
    exa.config.mode = 'exact'
    with torch.no_grad():
        color, radii, depth, alpha = exa.GaussianRasterizer(st)(
            means3D=a['mean_3d'], means2D=torch.zeros_like(a['mean_3d']), opacities=a['opacity'], colors_precomp=a['rgb'],
            scales=a['scale'], rotations=a['rotation'])
    need = exa.required_capacity(a['mean_3d'], a['opacity'], a['scale'], a['rotation'], colors_precomp=a['rgb'], settings=st)
    assert need > 64
    with exa.StaticRender(a['mean_3d'], a['opacity'], a['scale'], a['rotation'], colors_precomp=a['rgb'], image_size=(H, W),
                          capacity=need, train=False) as sr:
        v = sr.add_view(st)
        sr.forward(v)
        sr.check()
        assert torch.equal(sr.color, color) and torch.equal(sr.depth, depth) and torch.equal(sr.alpha, alpha)
        with pytest.raises(RuntimeError, match='not a training render'):
            sr.backward()
    with exa.StaticRender(a['mean_3d'], a['opacity'], a['scale'], a['rotation'], colors_precomp=a['rgb'], image_size=(H, W),
                          capacity=max(64, need // 4), train=False) as sr:
        v = sr.add_view(st)
        sr.forward(v)
        with pytest.raises(RuntimeError, match='needed %d instances' % need):
            sr.check()
        sr.forward(v)                                   # the object stays usable (and keeps raising for this camera)
        with pytest.raises(RuntimeError, match='needed'):
            sr.check()


def test_static_render_reads_its_cameras_in_place(dev):
    """The settings of a view point INTO a resident table: rewriting the table row changes what the next call renders."""
    H, W, f, P = 96, 112, 160.0, 3000
    a = {k: v.to(dev) for k, v in scenes.dist_b_avatar(P, seed=9).items()}
    bg = torch.zeros(3, device=dev)
    rows = []
    for v in (2, 11):
        tanx, tany, view, proj, campos = make_raster_matrices(scenes.ring_camera(H, W, v, 24, focal=f), (H, W))
        rows.append(torch.cat((view.reshape(-1), proj.reshape(-1), campos.reshape(-1), torch.zeros(13))))
    tab = torch.stack(rows).to(dev)
    slot = tab[0].clone()
    st = exa.GaussianRasterizationSettings(H, W, tanx, tany, bg, 1.0, slot[0:16].view(4, 4), slot[16:32].view(4, 4), 0, slot[32:35],
                                           False, False)
    exa.config.mode = 'exact'
    imgs = []
    with exa.StaticRender(a['mean_3d'], a['opacity'], a['scale'], a['rotation'], colors_precomp=a['rgb'], image_size=(H, W),
                          capacity=exa.required_capacity(a['mean_3d'], a['opacity'], a['scale'], a['rotation'], colors_precomp=a['rgb'],
                                                         settings=st) * 2, train=False) as sr:
        v = sr.add_view(st)
        for r in (0, 1):
            slot.copy_(tab[r])
            sr.forward(v)
            sr.check()
            imgs.append(sr.color.clone())
    assert not torch.equal(imgs[0], imgs[1])
    for r in (0, 1):
        str_ = exa.GaussianRasterizationSettings(H, W, tanx, tany, bg, 1.0, tab[r, 0:16].view(4, 4), tab[r, 16:32].view(4, 4), 0,
                                                 tab[r, 32:35], False, False)
        with torch.no_grad():
            ref = exa.GaussianRasterizer(str_)(means3D=a['mean_3d'], means2D=torch.zeros_like(a['mean_3d']), opacities=a['opacity'],
                                               colors_precomp=a['rgb'], scales=a['scale'], rotations=a['rotation'])[0]
        assert torch.equal(imgs[r], ref)
