"""Developer tool: per-kernel HIP-event times of K identical views of config C3 in ONE batched launch (eager), K = 1, 2, 4:
how each kernel scales when the chip holds K times the waves -- a kernel whose time barely grows is bound by per-wave
latency at low occupancy, not by throughput.  python tools/gpu_kernel_times_k.py [view]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes, _lib
from exavatar_release_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians_batch
from exavatar_release_amd.camera import make_raster_matrices

dev = torch.device('cuda:0')
H = W = 1024
P = 150000
view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
assets = scenes.dist_b_avatar(P, seed=0)
params = [assets[k].to(dev).requires_grad_(True) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
G = torch.randn(3, H, W, device=dev)
tanx, tany, vm, pm, cpos = make_raster_matrices(scenes.ring_camera(H, W, view, 200), (H, W))
st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, vm.to(dev), pm.to(dev), 0, cpos.to(dev), False, False)
exa.config.mode = 'exact'
for K in (1, 2, 3, 4, 8):
    m2 = [torch.zeros(P, 3, device=dev, requires_grad=True) for _ in range(K)]
    acc = {}
    for rep in range(5):
        _lib.timing_enable(rep > 1)
        m3, sc, rot, op, rgb = params
        jobs = [dict(means3D=m3, means2D=m2[k], shs=None, colors_precomp=rgb, opacities=op, scales=sc, rotations=rot,
                     cov3D_precomp=None, raster_settings=st) for k in range(K)]
        outs = rasterize_gaussians_batch(jobs)
        torch.autograd.grad([o[0] for o in outs], params + m2, grad_outputs=[G] * K)
        torch.cuda.synchronize()
        if rep > 1:
            for n, v in _lib.timing_read().items():
                acc[n] = acc.get(n, 0.0) + v / 3 * 1e3
    print('K=%d  ' % K + '  '.join('%s=%.1f' % (kk, vv) for kk, vv in acc.items()) + '  total=%.1f us  per view %.1f' % (sum(acc.values()), sum(acc.values()) / K))
