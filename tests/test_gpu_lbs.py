"""BASELINE.json configs[2] words the headline workload as "~150k Gaussians + SMPL-X LBS": the rasterizer's inputs are
NON-LEAF tensors produced by linear blend skinning (reference avatar/common/nets/module.py:413-422, 516-586), and its
gradients must flow on into pose / translation / offset parameters.  ``exavatar_release_amd.lbs.SyntheticAvatar`` is the
PyTorch-ROCm front end (the LBS stays in PyTorch, north_star); here: dL/dpose etc. through the HIP rasterizer against the
same module in front of the CPU oracle, and the same through ``GraphedIteration``."""
import copy

import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import lbs, scenes
from oracle import raster_oracle as ro
from tests.helpers import assert_grads_close, assert_image_close, gaussians_near_pixels

pytestmark = pytest.mark.gpu

H, W, F = 192, 160, 270.0
PARAMS = ('pose', 'trans', 'mean_offset', 'scale_log', 'rgb_logit')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    from exavatar_release_amd import _lib
    _lib.load()
    exa.config.mode, exa.config.fixed_capacity = 'exact', None
    return torch.device('cuda:0')


def _avatar(P, seed):
    base = scenes.dist_b_avatar(P, seed=seed)
    base['scale'] = base['scale'] * 2.0            # ~1 px splats at this focal length
    m = lbs.SyntheticAvatar(base, seed=seed)
    with torch.no_grad():                            # move away from the rest pose: the skinning must matter
        m.pose += 0.08 * torch.randn(m.pose.shape, generator=torch.Generator().manual_seed(seed + 1))
        m.trans += torch.tensor([0.02, -0.03, 0.05])
    return m


def test_gradients_reach_pose_and_offsets_through_the_rasterizer(dev):
    cam = scenes.neutral_camera(H, W, focal=F)
    bg = torch.tensor([0.2, 0.5, 0.8])
    G = torch.randn(3, H, W, generator=torch.Generator().manual_seed(3))
    m_cpu = _avatar(3000, 5)
    m_gpu = copy.deepcopy(m_cpu).to(dev)
    a = m_gpu()
    assert not a['mean_3d'].is_leaf and not a['scale'].is_leaf and not a['rgb'].is_leaf
    out = exa.GaussianRenderer()(a, (H, W), {k: v.to(dev) for k, v in cam.items()}, bg.to(dev))
    (out['img'] * G.to(dev)).sum().backward()
    a_ref = m_cpu()
    ref = ro.render(a_ref, (H, W), cam, bg, return_aux=True)
    (ref['img'] * G).sum().backward()
    amb = ro.ambiguous_pixel_mask(ref['aux'], H, W)
    assert_image_close(out['img'], ref['img'], amb, 'img')
    assert int(out['is_vis'].sum()) > 2500
    near = gaussians_near_pixels(ref['aux']['pre'], amb)
    for name in PARAMS:
        g, r = getattr(m_gpu, name).grad, getattr(m_cpu, name).grad
        assert g is not None and float(r.abs().max()) > 0, name
        per = name in ('mean_offset', 'scale_log', 'rgb_logit')
        assert_grads_close(g.reshape(r.shape), r, name, near if per else None, per_gaussian=per)


def test_lbs_outputs_through_the_graphed_iteration(dev):
    """The five-render iteration with the human sets produced by LBS (non-leaf), eager and replayed from hipGraphs:
    same images, and the gradients that reach pose / offsets are the same to rounding."""
    g = torch.Generator().manual_seed(8)
    scene = {k: v.to(dev) for k, v in scenes.dist_a_random(2500, H, W, seed=9, focal=F, z_range=(3.5, 6.0)).items()}
    cam = {k: v.to(dev) for k, v in scenes.neutral_camera(H, W, focal=F).items()}
    bg = torch.rand(3, generator=g).to(dev)
    G = [torch.randn(3, H, W, generator=g).to(dev) for _ in range(5)]
    m0 = _avatar(2000, 6)
    res = []
    with exa.GraphedIteration((H, W), dev) as it:
        for how in ('eager', 'graphed', 'graphed'):
            m = copy.deepcopy(m0).to(dev)
            human = m()
            refined = dict(human)
            refined['mean_3d'] = human['mean_3d'] + 0.004 * torch.sin(17.0 * m.mean_offset)        # a pose-dependent refinement
            s = {k: v.detach().clone().requires_grad_(True) for k, v in scene.items()}
            if how == 'eager':
                out = exa.render_iteration(exa.GaussianRenderer(), s, human, refined, (H, W), cam, bg)
            else:
                out = it(s, human, refined, cam, bg)
            imgs = [out[k]['img'].detach().clone() for k in exa.ITERATION_RENDERS]
            sum((out[k]['img'] * G[i]).sum() for i, k in enumerate(exa.ITERATION_RENDERS)).backward()
            torch.cuda.synchronize()
            res.append((imgs, [getattr(m, n).grad.clone() for n in PARAMS] + [s['mean_3d'].grad.clone()]))
        assert it.captures == 1
    for imgs, grads in res[1:]:
        for x, y in zip(imgs, res[0][0]):
            assert torch.equal(x, y)
        for x, y, n in zip(grads, res[0][1], PARAMS + ('scene mean_3d',)):
            scale = float(y.abs().max())
            assert scale > 0 and float((x - y).abs().max()) <= 1e-5 * scale, n
