"""Developer tool: per-kernel HIP-event times for a few ring views of config C3 (eager launches)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes, _lib
from exavatar_release_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians, last_header, _debug_last
from exavatar_release_amd.camera import make_raster_matrices

dev = torch.device('cuda:0')
H = W = 1024
P = int(os.environ.get('P', 150000))
assets = scenes.dist_b_avatar(P, seed=0)
params = [assets[k].to(dev).requires_grad_(True) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
mean_2d = torch.zeros(P, 3, device=dev, requires_grad=True)
G = torch.randn(3, H, W, device=dev)
views = [int(v) for v in (sys.argv[1:] or [0, 25, 50, 75, 100])]
exa.config.mode = 'exact'
exa.config.keep_debug = True
for k in views:
    tanx, tany, view, proj, cpos = make_raster_matrices(scenes.ring_camera(H, W, k, 200), (H, W))
    st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0, cpos.to(dev), False, False)
    acc = {}
    for rep in range(4):
        _lib.timing_enable(rep > 0)
        m3, sc, rot, op, rgb = params
        color, radii, depth, alpha = rasterize_gaussians(m3, mean_2d, None, rgb, op, sc, rot, None, st)
        grads = torch.autograd.grad([color], params + [mean_2d], grad_outputs=[G])
        torch.cuda.synchronize()
        if rep > 0:
            for n, v in _lib.timing_read().items():
                acc[n] = acc.get(n, 0.0) + v / 3 * 1e3
    hdr = last_header()
    print('view %3d slots*64=%d entries=%d' % (k, hdr[0], hdr[2]))
    print('   ' + '  '.join('%s=%.1f' % (kk, vv) for kk, vv in acc.items()) + '  total=%.1f us' % sum(acc.values()))
