"""The five-render ExAvatar iteration (bench.py extra_exavatar_iteration, 'sets' flavour) in a loop, for
`rocprofv3 --kernel-trace --stats`: per-kernel GPU time of one iteration.  Usage: python tools/gpu_iteration_profile.py [how] [iters]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
how = sys.argv[1] if len(sys.argv) > 1 else 'sets'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device('cuda:0'); H = W = 1024
keys = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')
scene = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_c_scene(100_000, H, W, seed=1).items()}
human = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_b_avatar(50_000, seed=2).items()}
refined = {k: v.detach().clone().requires_grad_(True) for k, v in human.items()}
cam = {k: t.to(dev) for k, t in scenes.ring_camera(H, W, 7, 200).items()}
bg = torch.rand(3, device=dev); G = torch.randn(3, H, W, device=dev)
rend = exa.GaussianRenderer()
cat = lambda a, b: {k: torch.cat((a[k].detach(), b[k])) for k in keys}      # noqa: E731


graphed = exa.GraphedIteration((H, W), dev) if how == 'graphed' else None
if how == 'graphed_loss':       # the same loss recorded into the graph: forward + loss + backward = one replay
    graphed = exa.GraphedIteration((H, W), dev, loss_fn=lambda out, G_: sum((out[k]['img'] * G_).sum() for k in exa.ITERATION_RENDERS))


if graphed is not None:
    graphed.tight_backward = os.environ.get('EXA_TIGHT', '1') != '0'      # A/B knob


def iteration():
    if how == 'graphed_loss':
        for t in (scene, human, refined):
            for v in t.values():
                v.grad = None
        graphed(scene, human, refined, cam, bg, loss_args=(G,))['loss'].backward()
        return
    if how == 'graphed':
        res = graphed(scene, human, refined, cam, bg)
        outs = [res[k] for k in exa.ITERATION_RENDERS]
    elif how == 'sets':
        res = exa.render_iteration(rend, scene, human, refined, (H, W), cam, bg)
        outs = [res[k] for k in exa.ITERATION_RENDERS]
    else:
        jobs = [(scene, (H, W), cam), (human, (H, W), cam, bg), (cat(scene, human), (H, W), cam),
                (refined, (H, W), cam, bg), (cat(scene, refined), (H, W), cam)]
        outs = exa.render_many(rend, jobs) if how == 'batched' else [rend(*j) for j in jobs]
    loss = sum((o['img'] * G).sum() for o in outs)
    for t in (scene, human, refined):
        for v in t.values():
            v.grad = None
    loss.backward()


exa.config.fold_composite_grads = os.environ.get('EXA_FOLD', '1') != '0'      # A/B knob
exa.config.overlap_composites = os.environ.get('EXA_OVERLAP', '1') != '0'    # A/B knob
exa.config.mode = 'exact'
for _ in range(2):
    iteration()
exa.config.mode = 'auto'
for _ in range(8):
    iteration()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters):
    iteration()
th = time.perf_counter() - t0
torch.cuda.synchronize()
print('%s: %.3f ms / iteration (host %.3f ms)' % (how, (time.perf_counter() - t0) / iters * 1e3, th / iters * 1e3))
