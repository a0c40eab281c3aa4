import sys, os, time, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
dev = torch.device('cuda:0'); H = W = 1024
human = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_b_avatar(150000, seed=2).items()}
cam = {k: t.to(dev) for k, t in scenes.ring_camera(H, W, 7, 200).items()}
bg = torch.ones(3, device=dev); G = torch.randn(3, H, W, device=dev)
rend = exa.GaussianRenderer()
exa.config.mode = os.environ.get('EXA_MODE', 'auto')
def it():
    o = rend(human, (H, W), cam, bg)
    for v in human.values(): v.grad = None
    (o['img'] * G).sum().backward()
for _ in range(10): it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): it()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print('eager render fwd+bwd through GaussianRenderer (mode %s): %.3f ms -> %.0f renders/s' % (exa.config.mode, dt * 1e3, 1 / dt))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): it()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
pstats.Stats(pr).sort_stats('tottime').print_stats(30)

# host time of the two halves (no synchronisation inside the loop: what the Python thread spends queueing work)
import statistics
tf, tb = [], []
for _ in range(60):
    for v in human.values(): v.grad = None
    t0 = time.perf_counter(); o = rend(human, (H, W), cam, bg); t1 = time.perf_counter()
    L = (o['img'] * G).sum(); t2 = time.perf_counter()
    L.backward(); t3 = time.perf_counter()
    tf.append(t1 - t0); tb.append(t3 - t2)
    if _ % 8 == 7: torch.cuda.synchronize()
print('host: forward %.1f us, backward (engine + raster backward + AccumulateGrad) %.1f us (medians)' % (statistics.median(tf) * 1e6, statistics.median(tb) * 1e6))
# the five-render iteration of bench.py (extra_exavatar_iteration), 'sets' flavour
import bench
print(bench.iteration_throughput(dev, iters=20))
