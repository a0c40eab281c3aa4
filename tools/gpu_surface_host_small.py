"""Pure host cost of the drop-in surface per fwd + bwd: a render so small (64 Gaussians, 64 x 64) that the device is always
waiting for the host.  Usage: python tools/gpu_surface_host_small.py"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes, rasterizer as rz
from exavatar_release_amd.camera import make_raster_matrices
dev = torch.device('cuda:0'); H = W = 64; P = 64
a = scenes.dist_a_random(P, H, W, seed=0, focal=100.0)
params = [a[k].to(dev).contiguous().requires_grad_(True) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
m3, sc, rot, op, col = params
G = torch.randn(3, H, W).to(dev)
bg = torch.ones(3, device=dev)
tanx, tany, view, proj, campos = make_raster_matrices(scenes.neutral_camera(H, W, focal=100.0), (H, W))
st = exa.GaussianRasterizationSettings(H, W, tanx, tany, bg, 1.0, view.to(dev), proj.to(dev), 0, campos.to(dev), False, False)
rast = exa.GaussianRasterizer(st)
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
ins = params + [m2]
exa.config.min_capacity = 64


def step():
    color, radii, depth, alpha = rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=col, scales=sc, rotations=rot)
    return torch.autograd.grad([color], ins, grad_outputs=[G])


def fwd_only():
    return rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=col, scales=sc, rotations=rot)


for how in ('off', 'auto'):
    exa.config.compiled_node = how
    for i in range(50):
        step()
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for i in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for i in range(n):
        fwd_only()
    torch.cuda.synchronize()
    df = (time.perf_counter() - t0) / n
    print('compiled_node=%-4s  fwd + bwd %.1f us/step, forward alone (context kept, dropped) %.1f us' % (how, dt * 1e6, df * 1e6), flush=True)
with torch.autograd.set_multithreading_enabled(False):
    for i in range(50):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2000):
        step()
    torch.cuda.synchronize()
    print('compiled node, autograd multithreading off (backward on the calling thread): fwd + bwd %.1f us/step' % ((time.perf_counter() - t0) / 2000 * 1e6), flush=True)
x = torch.randn(64, device=dev, requires_grad=True)
gx = torch.ones(64, device=dev)
for mt in (True, False):
    with torch.autograd.set_multithreading_enabled(mt):
        for i in range(50):
            torch.autograd.grad([x * 2.0], [x], grad_outputs=[gx])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2000):
            torch.autograd.grad([x * 2.0], [x], grad_outputs=[gx])
        torch.cuda.synchronize()
        print('torch baseline: y = x * 2 on the device + autograd.grad, multithreading %s: %.1f us/step' % (mt, (time.perf_counter() - t0) / 2000 * 1e6), flush=True)
pr = cProfile.Profile(); pr.enable()
for i in range(2000):
    step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
