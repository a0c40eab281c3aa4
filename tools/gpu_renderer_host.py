"""Host cost of `exa.GaussianRenderer.forward` (the mirror of reference module.py:592-647) on top of the rasterizer call it wraps:
a render so small that the device always waits.  python tools/gpu_renderer_host.py"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes, renderer
dev = torch.device('cuda:0'); H = W = 64; P = 64
a = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_a_random(P, H, W, seed=0, focal=100.0).items()}
cam = {k: v.to(dev) for k, v in scenes.neutral_camera(H, W, focal=100.0).items()}
bg = torch.ones(3, device=dev); G = torch.randn(3, H, W, device=dev)
rend = exa.GaussianRenderer()
exa.config.min_capacity = 64
job = renderer._raster_job(a, (H, W), cam, bg)
rast = exa.GaussianRasterizer(job['raster_settings'])
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)


def via_renderer():
    out = rend(a, (H, W), cam, bg)
    torch.autograd.backward([out['img']], [G])


def via_rasterizer():
    color = rast(means3D=a['mean_3d'], means2D=m2, opacities=a['opacity'], colors_precomp=a['rgb'], scales=a['scale'], rotations=a['rotation'])[0]
    torch.autograd.backward([color], [G])


def fwd_renderer():
    return rend(a, (H, W), cam, bg)


def fwd_rasterizer():
    return rast(means3D=a['mean_3d'], means2D=m2, opacities=a['opacity'], colors_precomp=a['rgb'], scales=a['scale'], rotations=a['rotation'])


for name, fn in (('renderer fwd+bwd', via_renderer), ('rasterizer fwd+bwd', via_rasterizer), ('renderer fwd', fwd_renderer), ('rasterizer fwd', fwd_rasterizer)) * 2:
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    torch.cuda.synchronize()
    print('%-20s %.1f us' % (name, (time.perf_counter() - t0) / 2000 * 1e6), flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(2000):
    fwd_renderer()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
