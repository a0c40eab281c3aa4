"""Developer tool: where the time of a GraphedIteration goes (host segments by perf_counter, device time by events), next
to the eager render_iteration, on bench.py's iteration workload (100 k Dist-C scene + 50 k avatar, 1024^2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes

dev = torch.device('cuda:0')
H = W = 1024
scene = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_c_scene(100_000, H, W, seed=1).items()}
human = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_b_avatar(50_000, seed=2).items()}
refined = {k: v.detach().clone().requires_grad_(True) for k, v in human.items()}
cam = {k: t.to(dev) for k, t in scenes.ring_camera(H, W, 7, 200).items()}
bg = torch.rand(3, device=dev)
G = torch.randn(3, H, W, device=dev)
rend = exa.GaussianRenderer()
exa.config.mode = 'auto'


def run(name, fwd, n=40, sync_between=False):
    seg = [0.0, 0.0, 0.0]
    for i in range(n + 10):
        if i == 10:
            torch.cuda.synchronize(); seg = [0.0, 0.0, 0.0]; t_all = time.perf_counter()
        for t in (scene, human, refined):
            for v in t.values():
                v.grad = None
        t0 = time.perf_counter()
        res = fwd()
        if sync_between: torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss = sum((res[k]['img'] * G).sum() for k in exa.ITERATION_RENDERS)
        if sync_between: torch.cuda.synchronize()
        t2 = time.perf_counter()
        loss.backward()
        if sync_between: torch.cuda.synchronize()
        t3 = time.perf_counter()
        seg[0] += t1 - t0; seg[1] += t2 - t1; seg[2] += t3 - t2
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t_all) / n * 1e3
    print('%-28s total %.3f ms/iter | forward %.3f  loss %.3f  backward %.3f (%s)' % (
        name, tot, seg[0] / n * 1e3, seg[1] / n * 1e3, seg[2] / n * 1e3, 'device-inclusive' if sync_between else 'host only'))


eager = lambda: exa.render_iteration(rend, scene, human, refined, (H, W), cam, bg)
for _ in range(3):
    exa.config.mode = 'exact'; r = eager(); sum((r[k]['img'] * G).sum() for k in exa.ITERATION_RENDERS).backward(); exa.config.mode = 'auto'
it = exa.GraphedIteration((H, W), dev)
it_nc = exa.GraphedIteration((H, W), dev, check=False)
graphed = lambda: it(scene, human, refined, cam, bg)
graphed_nc = lambda: it_nc(scene, human, refined, cam, bg)
for sync in (False, True):
    run('eager render_iteration', eager, sync_between=sync)
    run('GraphedIteration', graphed, sync_between=sync)
    run('GraphedIteration check=False', graphed_nc, sync_between=sync)
# bare replays of the two graphs (no input copies, no autograd)
cap = it._cap
g_b = next(iter(cap.bwd.values()))[0]          # the backward graph of the gradient pattern the loop above used
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    cap.fwd.replay(); g_b.replay()
torch.cuda.synchronize()
print('bare replay of both graphs: %.3f ms/iter' % ((time.perf_counter() - t0) / 50 * 1e3))
t0 = time.perf_counter()
for _ in range(50):
    cap.fwd.replay()
torch.cuda.synchronize()
print('bare replay forward graph: %.3f ms' % ((time.perf_counter() - t0) / 50 * 1e3))
t0 = time.perf_counter()
for _ in range(50):
    g_b.replay()
torch.cuda.synchronize()
print('bare replay backward graph: %.3f ms' % ((time.perf_counter() - t0) / 50 * 1e3))
