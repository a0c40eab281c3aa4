"""C5 (300 k Gaussians, SH degree 3, 2048 x 2048, forward only) per frame: eager no_grad GaussianRasterizer vs the
product's GraphedRenderer (hipGraph replay, new camera every frame) with and without the per-frame header check."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.camera import make_raster_matrices

dev = torch.device('cuda:0')
assets, shape, cam0 = scenes.make_config('c5'); H, W = shape
P = assets['mean_3d'].shape[0]
sh = scenes.sh_from_rgb(assets['rgb'], 3, seed=5, rest_sigma=0.1)
a = {k: assets[k].to(dev) for k in ('mean_3d', 'scale', 'rotation', 'opacity')}; a['sh'] = sh.to(dev)
cams = [{k: t.to(dev) for k, t in scenes.ring_camera(H, W, (k * 25) // 2, 200, focal=1500.0 * H / 1024).items()} for k in range(16)]   # 16 views spread over the ring, like bench.py's 200
bg = torch.ones(3, device=dev)
m2 = torch.zeros(P, 3, device=dev)


def eager(cam):
    tanx, tany, view, proj, campos = make_raster_matrices({k: v.cpu() for k, v in cam.items()}, shape)
    st = exa.GaussianRasterizationSettings(H, W, tanx, tany, bg, 1.0, view.to(dev), proj.to(dev), 3, campos.to(dev), False, False)
    with torch.no_grad():
        return exa.GaussianRasterizer(st)(means3D=a['mean_3d'], means2D=m2, opacities=a['opacity'], shs=a['sh'],
                                          scales=a['scale'], rotations=a['rotation'])


def timeit(fn, n=48):
    for i in range(8):
        fn(cams[i % 16])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fn(cams[i % 16])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print('eager no_grad GaussianRasterizer      : %.3f ms / frame' % timeit(eager))
for check in (True, False):
    gr = exa.GraphedRenderer(P, shape, dev, sh_degree=3, check=check)
    ms = timeit(lambda c: gr(a, c, bg))
    print('GraphedRenderer (check=%-5s)          : %.3f ms / frame = %.0f frames/s, %d capture(s)' % (check, ms, 1e3 / ms, gr.captures))
# animate.py's case: ONE camera for the whole sequence (camera block memoised), the Gaussians change every frame
gr = exa.GraphedRenderer(P, shape, dev, sh_degree=3, check=False)
frames = [dict(a, mean_3d=a['mean_3d'] + 0.001 * i) for i in range(4)]
ms = timeit(lambda c: gr(frames[id(c) % 4], cams[0], bg))
print('GraphedRenderer, fixed camera, new Gaussians every frame (check=False): %.3f ms / frame = %.0f frames/s' % (ms, 1e3 / ms))
rgb_in = {k: a[k] for k in ('mean_3d', 'scale', 'rotation', 'opacity')}; rgb_in['rgb'] = assets['rgb'].to(dev)
gr2 = exa.GraphedRenderer(P, shape, dev, check=False)
ms = timeit(lambda c: gr2(rgb_in, cams[0], bg))
print('... with precomputed colours instead of SH (the reference\'s own path)  : %.3f ms / frame = %.0f frames/s' % (ms, 1e3 / ms))

# where a frame's time goes: the bare replay of the captured graph, + the camera kernel, the whole __call__
g = gr2._graph
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(64):
    g.replay()
torch.cuda.synchronize(); print('bare graph replay (rgb)              : %.3f ms' % ((time.perf_counter() - t0) / 64 * 1e3))
from exavatar_release_amd.renderer import camera_block_device
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(64):
    camera_block_device(cams[i % 16], shape, gr2._cam)
    g.replay()
torch.cuda.synchronize(); print('camera kernel (host path memo) + replay: %.3f ms' % ((time.perf_counter() - t0) / 64 * 1e3))
t0 = time.perf_counter()
for i in range(64):
    gr2(rgb_in, cams[i % 16], bg)
t1 = time.perf_counter(); torch.cuda.synchronize()
print('GraphedRenderer.__call__ (rgb, new camera / frame, check=False): host %.3f ms / frame, wall %.3f ms / frame' % ((t1 - t0) / 64 * 1e3, (time.perf_counter() - t0) / 64 * 1e3))
gr3 = exa.GraphedRenderer(P, shape, dev, check=True)
ms = timeit(lambda c: gr3(rgb_in, c, bg))
print('GraphedRenderer (rgb, new camera / frame, check=True): %.3f ms / frame' % ms)
