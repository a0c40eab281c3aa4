import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.rasterizer import last_header
dev = torch.device('cuda:0')
H, W = 2160, 3840
P = 2_000_000
a = scenes.cat_assets(scenes.dist_c_scene(500_000, H, W, 0), scenes.dist_b_avatar(P - 500_000, 0))
a = {k: v.to(dev).requires_grad_(True) for k, v in a.items()}
cam = {k: t.to(dev) for k, t in scenes.neutral_camera(H, W).items()}
exa.config.mode = 'exact'
rend = exa.GaussianRenderer()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = rend(a, (H, W), cam)
    out['img'].mean().backward()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('4K, P=2M: fwd+bwd %.2f ms; header (capacity needed, overflow, entries, visible, instances) = %s; img mean %.4f finite %s grad finite %s' % (
        dt * 1e3, last_header(), float(out['img'].mean()), bool(torch.isfinite(out['img']).all()), bool(torch.isfinite(a['mean_3d'].grad).all())))
    for v in a.values(): v.grad = None
print('max mem GB', torch.cuda.max_memory_allocated() / 1e9)
