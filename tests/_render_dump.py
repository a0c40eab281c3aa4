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
# a stack of 700 faint splats over one 8x8 sub-tile: a list of more than 512 keys (the "long list" paths of the sort)
g0 = torch.Generator().manual_seed(11)
n_deep = 700
depth = 2.0 + 4.0 * torch.rand(n_deep, generator=g0)
for axis, centre in ((0, 100.0 - W / 2 + 0.5), (1, 60.0 - H / 2 + 0.5)):
    assets['mean_3d'][:n_deep, axis] = (centre + (torch.rand(n_deep, generator=g0) - 0.5) * 5.0) / 560.0 * depth
assets['mean_3d'][:n_deep, 2] = depth
assets['scale'][:n_deep] = (0.003 + 0.003 * torch.rand(n_deep, 3, generator=g0)) * depth.view(-1, 1)
assets['opacity'][:n_deep] = 0.006 + 0.004 * torch.rand(n_deep, 1, generator=g0)
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
