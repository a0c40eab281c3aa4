"""Measures SURVEY 8f-2 / 8f-3 on one GPU: the five same-camera renders of one ExAvatar iteration (fwd + bwd)
issued sequentially vs as ONE batched call (exa.render_many), and the fused densify statistics vs the reference's PyTorch bookkeeping."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from oracle import raster_oracle as ro   # densify_stats_reference only (the PyTorch statements of the reference)

dev = torch.device('cuda:0'); H = W = 1024
KEYS = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')
if os.environ.get('EXA_SCENE', 'avatar') == 'scene':      # a real background (Dist-C) instead of a second body
    scene = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_c_scene(100000, H, W, seed=1).items()}
else:
    scene = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_b_avatar(100000, seed=1).items()}
human = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_b_avatar(50000, seed=2).items()}
refined = {k: v.detach().clone().requires_grad_(True) for k, v in human.items()}
cam = {k: t.to(dev) for k, t in scenes.ring_camera(H, W, 7, 200).items()}
bg = torch.ones(3, device=dev)
G = torch.randn(3, H, W, device=dev)
cat = lambda a, b: {k: torch.cat((a[k].detach(), b[k])) for k in KEYS}
rend = exa.GaussianRenderer()


def iteration(concurrent):
    if concurrent == 'sets':          # Gaussian sets shared between the renders, constant scene prefix in the composites
        res = exa.render_iteration(rend, scene, human, refined, (H, W), cam, bg)
        outs = [res[k] for k in exa.ITERATION_RENDERS]
    else:
        jobs = [(scene, (H, W), cam), (human, (H, W), cam, bg), (cat(scene, human), (H, W), cam), (refined, (H, W), cam, bg),
                (cat(scene, refined), (H, W), cam)]
        outs = exa.render_many(rend, jobs) if concurrent else [rend(*j) for j in jobs]
    loss = sum((o['img'] * G).sum() for o in outs)
    for t in (scene, human, refined):
        for v in t.values():
            v.grad = None
    loss.backward()
    return outs


for mode in ('exact', 'auto'):
    exa.config.mode = mode
    for conc in (False, True, 'sets'):
        for _ in range(60):          # long warm-up: the first phase after a protocol switch measured slow for dozens of iterations
            iteration(conc)
            torch.cuda.synchronize()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            iteration(conc)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print('5 renders fwd+bwd, mode %-8s %-10s: %.3f ms / iteration' % (mode, {False: 'sequential', True: 'batched', 'sets': 'sets'}[conc], dt * 1e3))

outs = iteration(False)
g2d, radius = outs[0]['mean_2d'].grad, outs[0]['radius']
P = radius.shape[0]
acc, cnt, rmax = torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev), torch.zeros(P, device=dev)
for name, fn in (('fused HIP kernel', lambda: exa.track_densify_stats(g2d, radius, acc, cnt, rmax)),
                 ('reference PyTorch statements', lambda: ro.densify_stats_reference(g2d, radius, acc, cnt, rmax))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    print('densify statistics of one render (P = %d), %-30s: %.1f us' % (P, name, (time.perf_counter() - t0) / 50 * 1e6))
