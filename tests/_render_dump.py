"""Helper of test_footprint_cull_leaves_the_result_alone (run as a subprocess so that the library reads EXA_FOOTPRINT
afresh): renders a small avatar + scene mixture fwd + bwd through the drop-in surface and saves every output."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exavatar_release_amd as exa                                          # noqa: E402
from exavatar_release_amd import scenes                                      # noqa: E402

out = sys.argv[1]
dev = torch.device('cuda:0')
H, W = 384, 512
assets = scenes.cat_assets(scenes.dist_c_scene(4000, H, W, 3), scenes.dist_b_avatar(20000, 3))
cam = scenes.neutral_camera(H, W, focal=560.0)
params = {k: assets[k].to(dev).requires_grad_(True) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')}
renderer = exa.GaussianRenderer()
res = renderer(params, (H, W), {k: v.to(dev) for k, v in cam.items()}, bg=torch.tensor([0.2, 0.5, 0.8], device=dev))
g = torch.Generator().manual_seed(5)
G = torch.randn(3, H, W, generator=g).to(dev)
Gd = torch.randn(1, H, W, generator=g).to(dev)
((res['img'] * G).sum() + (res['depthmap'] * Gd).sum() + (res['mask'] * Gd).sum() * 0.5).backward()
arrays = {'img': res['img'], 'depth': res['depthmap'], 'mask': res['mask'], 'radius': res['radius'],
          'mean_2d_grad': res['mean_2d'].grad}
arrays.update({'grad_' + k: v.grad for k, v in params.items()})
np.savez(out, **{k: v.detach().cpu().numpy() for k, v in arrays.items()})
