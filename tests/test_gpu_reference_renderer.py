"""The HIP path behind ``exa.GaussianRenderer`` against ``tests/golden/ref_renderer.npz``: what the reference's own
``class GaussianRenderer`` (module.py:588-647, exec'd unchanged by tests/golden/make_golden_renderer.py) returned for the same
inputs with the CPU oracle as its rasterizer -- the six-key dict, ``mean_2d.grad`` (the densification signal of
avatar/main/train.py:51) and the asset gradients, for a C1-like scene and a 540-pixel-wide avatar view."""
import os

import numpy as np
import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import rasterizer as rz
from tests import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['c1', 'w540'])
def test_hip_renderer_matches_the_reference_callers_outputs(golden_dir, name):
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    dev = torch.device('cuda:0')
    gold = np.load(os.path.join(golden_dir, 'ref_renderer.npz'))
    p = name + '_'
    assets = {k[len(p) + 6:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(p + 'asset_')}
    cam = {k[len(p) + 4:]: torch.from_numpy(gold[k]).to(dev) for k in gold.files if k.startswith(p + 'cam_')}
    H, W = (int(v) for v in gold[p + 'shape'])
    bg = torch.from_numpy(gold[p + 'bg']).to(dev) if bool(gold[p + 'bg_given']) else None
    G = torch.from_numpy(gold[p + 'G']).to(dev)
    amb = torch.from_numpy(gold[p + 'ambiguous'])
    rend = exa.GaussianRenderer()
    for rep in range(2):                      # first call of a shape: Python node (exact); second: the compiled node
        a = {k: v.to(dev).requires_grad_(True) for k, v in assets.items()}
        n0 = rz.compiled_calls
        out = rend(a, (H, W), cam) if bg is None else rend(a, (H, W), cam, bg)
        assert sorted(out) == ['depthmap', 'img', 'is_vis', 'mask', 'mean_2d', 'radius']
        (out['img'] * G).sum().backward()
        if rep:
            assert rz.compiled_calls == n0 + 1, (rz._compiled.last_decline() if rz._compiled else rz._compiled, exa.config.__dict__, rz._capture_report)
        assert torch.equal(out['radius'].cpu(), torch.from_numpy(gold[p + 'radius']))          # int32, bit-equal
        assert out['radius'].dtype == torch.int32 and out['is_vis'].dtype == torch.bool
        assert torch.equal(out['is_vis'].cpu(), torch.from_numpy(gold[p + 'is_vis']))
        assert tuple(out['img'].shape) == (3, H, W) and tuple(out['depthmap'].shape) == (1, H, W) and tuple(out['mask'].shape) == (1, H, W)
        for k in ('img', 'depthmap', 'mask'):
            st = hp.assert_image_close(out[k], torch.from_numpy(gold[p + k]), amb, name=k)
            hp.record_stats('ref_renderer_%s_%s' % (name, k), st)
        assert out['mean_2d'].is_leaf and out['mean_2d'].grad is not None
        scale_ref = torch.from_numpy(gold[p + 'grad_scale'])
        hp.assert_grads_close(out['mean_2d'].grad, torch.from_numpy(gold[p + 'mean_2d_grad']), 'mean_2d', per_gaussian=False)
        assert float(out['mean_2d'].grad[:, 2].abs().max()) == 0.0
        for k in assets:
            ref = torch.from_numpy(gold[p + 'grad_' + k])
            hp.assert_grads_close(a[k].grad, ref, k, per_gaussian=False,
                                  abs_scale=hp.rotation_grad_scale(assets['scale'], scale_ref) if k == 'rotation' else 0.0)
