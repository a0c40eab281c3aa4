"""What a (re-)capture of GraphedIteration costs on the host: first call, a change of P (densification), steady state.
100 k Dist-C scene + 50 k avatar, 1024 x 1024.  Usage: python tools/gpu_capture_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
dev = torch.device('cuda:0'); H = W = 1024
mk = lambda d: {k: v.to(dev).requires_grad_(True) for k, v in d.items()}
human = mk(scenes.dist_b_avatar(50_000, seed=2)); refined = {k: v.detach().clone().requires_grad_(True) for k, v in human.items()}
cam = {k: t.to(dev) for k, t in scenes.ring_camera(H, W, 7, 200).items()}
bg = torch.rand(3, device=dev); G = torch.randn(3, H, W, device=dev)
it = exa.GraphedIteration((H, W), dev)


phases = []


def step(scene):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = it(scene, human, refined, cam, bg)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss = sum((res[k]['img'] * G).sum() for k in exa.ITERATION_RENDERS)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    phases[:] = [(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, it.overflow_retries]
    return (t3 - t0) * 1e3


for P in (100_000, 100_000, 100_000, 104_000, 104_000, 104_000, 98_000, 98_000, 98_000, 98_000):
    scene = mk(scenes.dist_c_scene(P, H, W, seed=1))
    t = step(scene)
    import gc; tg = time.perf_counter(); n_gc = gc.collect(); torch.cuda.synchronize(); tg = (time.perf_counter() - tg) * 1e3
    print('P_scene %6d: %8.2f ms = call %.2f + loss %.2f + backward %.2f (captures so far: forward %d, backward %d; overflow retries %d) | gc.collect() afterwards: %.2f ms, %d objects' % (P, t, phases[0], phases[1], phases[2], it.captures, it.backward_captures, phases[3], tg, n_gc))
