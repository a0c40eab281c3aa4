"""Developer probe: where the time of 5 sequential renders (fwd + bwd) goes in auto / capacity vs exact mode."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
dev = torch.device('cuda:0'); H = W = 1024
KEYS = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')
scene = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_b_avatar(100000, seed=1).items()}
human = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_b_avatar(50000, seed=2).items()}
cam = {k: t.to(dev) for k, t in scenes.ring_camera(H, W, 7, 200).items()}
bg = torch.ones(3, device=dev); G = torch.randn(3, H, W, device=dev)
rend = exa.GaussianRenderer()
def iteration(n_jobs=5):
    jobs = [(scene, (H, W), cam, bg), (human, (H, W), cam, bg)] * 3
    outs = [rend(*j) for j in jobs[:n_jobs]]
    loss = sum((o['img'] * G).sum() for o in outs)
    for t in (scene, human):
        for v in t.values(): v.grad = None
    loss.backward()
for mode, growth in (('exact', 1.5), ('auto', 1.5), ('auto', 1.0), ('capacity', 1.5)):
    exa.config.mode = mode; exa.config.capacity_growth = growth
    for _ in range(5): iteration()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): iteration()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print('mode %-8s growth %.1f: %.3f ms / iteration' % (mode, growth, dt * 1e3))
exa.config.mode = 'auto'; exa.config.capacity_growth = 1.5
pr = cProfile.Profile(); pr.enable()
for _ in range(30): iteration()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)

# transient after switching from the two-stage (exact) to the fused (capacity) protocol: per-iteration wall time
exa.config.mode = 'exact'
for _ in range(10): iteration()
torch.cuda.synchronize()
exa.config.mode = 'auto'; exa.config.capacity_growth = 1.7       # sizes no earlier phase has used
ts = []
for i in range(80):
    t0 = time.perf_counter(); iteration(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('per-iteration ms after the switch (synchronised each iteration):', ' '.join('%.2f' % t for t in ts[:12]), '...',
      ' '.join('%.2f' % t for t in ts[-6:]))
print('allocator: reserved %.1f GB, num_alloc_retries %d' % (torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_stats().get('num_alloc_retries', 0)))
