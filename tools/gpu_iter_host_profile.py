"""cProfile of the eager five-render iteration on the host: where the Python time goes.  `sets` = render_iteration + backward;
`sequential` = the reference's formulation, five GaussianRenderer calls on torch.cat((scene.detach(), human)) (model.py:119-167).
Usage: python tools/gpu_iter_host_profile.py [iters] [sets|sequential] [noprofile]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
how = sys.argv[2] if len(sys.argv) > 2 else 'sets'
dev = torch.device('cuda:0'); H = W = 1024
scene = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_c_scene(100_000, H, W, seed=1).items()}
human = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_b_avatar(50_000, seed=2).items()}
refined = {k: v.detach().clone().requires_grad_(True) for k, v in human.items()}
cam = {k: t.to(dev) for k, t in scenes.ring_camera(H, W, 7, 200).items()}
bg = torch.rand(3, device=dev); G = torch.randn(3, H, W, device=dev)
rend = exa.GaussianRenderer()


keys = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')
cat = lambda a, b: {k: torch.cat((a[k].detach(), b[k])) for k in keys}      # noqa: E731


def it():
    if how == 'sequential':
        jobs = [(scene, (H, W), cam), (human, (H, W), cam, bg), (cat(scene, human), (H, W), cam),
                (refined, (H, W), cam, bg), (cat(scene, refined), (H, W), cam)]
        imgs = [rend(*j)['img'] for j in jobs]
    else:
        res = exa.render_iteration(rend, scene, human, refined, (H, W), cam, bg)
        imgs = [res[k]['img'] for k in exa.ITERATION_RENDERS]
    for t in (scene, human, refined):
        for v in t.values():
            v.grad = None
    torch.autograd.backward(imgs, [G] * 5)


exa.config.mode = 'exact'
for _ in range(2):
    it()
exa.config.mode = 'auto'
for _ in range(20):
    it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters):
    it()
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(how + ': eager iteration (dL/dimg handed to backward): %.3f ms wall, host %.3f ms' % ((time.perf_counter() - t0) / iters * 1e3, th / iters * 1e3))
if 'noprofile' in sys.argv:
    sys.exit(0)
pr = cProfile.Profile(); pr.enable()
for _ in range(iters):
    it()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
