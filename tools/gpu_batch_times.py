"""Developer tool: per-kernel HIP-event times of ONE batched launch of K ring views of config C3 (eager), K = 1, 2, 4, 8:
what each pipeline stage costs when K views share a launch (exa_raster_forward_batch / _backward_batch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes, _lib
from exavatar_release_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians_batch
from exavatar_release_amd.camera import make_raster_matrices

dev = torch.device('cuda:0')
H = W = 1024
P = int(os.environ.get('P', 150000))
assets = scenes.dist_b_avatar(P, seed=0)
params = [assets[k].to(dev).requires_grad_(True) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
G = torch.randn(3, H, W, device=dev)
exa.config.mode = 'exact'
for K in [int(v) for v in (sys.argv[1:] or [1, 2, 4, 8])]:
    sts, m2s = [], []
    for k in range(K):
        tanx, tany, view, proj, cpos = make_raster_matrices(scenes.ring_camera(H, W, 25 * k, 200), (H, W))
        sts.append(GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0,
                                                 cpos.to(dev), False, False))
        m2s.append(torch.zeros(P, 3, device=dev, requires_grad=True))
    m3, sc, rot, op, rgb = params
    acc = {}
    for rep in range(4):
        _lib.timing_enable(rep > 0)
        jobs = [dict(means3D=m3, means2D=m2s[k], shs=None, colors_precomp=rgb, opacities=op, scales=sc, rotations=rot,
                     cov3D_precomp=None, raster_settings=sts[k]) for k in range(K)]
        outs = rasterize_gaussians_batch(jobs)
        torch.autograd.grad([o[0] for o in outs], params + m2s, grad_outputs=[G] * K)
        torch.cuda.synchronize()
        if rep > 0:
            for n, v in _lib.timing_read().items():
                acc[n] = acc.get(n, 0.0) + v / 3 * 1e3
    _lib.timing_enable(False)
    tot = sum(acc.values())
    print('K=%d  ' % K + '  '.join('%s=%.1f' % (kk, vv) for kk, vv in acc.items()) + '  total=%.1f us  (%.1f us / view)' % (tot, tot / K))
