"""3 000 fwd + bwd steps through the compiled autograd node: device memory (allocated / reserved) and the number of live Python objects
must be what they were after the warm-up.  python tools/gpu_leak_check.py"""
import os, sys, gc
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes, rasterizer as rz
from exavatar_release_amd.camera import make_raster_matrices
dev = torch.device('cuda:0'); H = W = 256; P = 20000
a = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_b_avatar(P, seed=0).items()}
tanx, tany, vm, pm, cp = make_raster_matrices(scenes.ring_camera(H, W, 3, 24, focal=400.0), (H, W))
st = exa.GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, vm.to(dev), pm.to(dev), 0, cp.to(dev), False, False)
G = torch.randn(3, H, W, device=dev)
rast = exa.GaussianRasterizer(st)
def step():
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    c = rast(means3D=a['mean_3d'], means2D=m2, opacities=a['opacity'], colors_precomp=a['rgb'], scales=a['scale'], rotations=a['rotation'])[0]
    (c * G).sum().backward()
    for v in a.values(): v.grad = None
for _ in range(50): step()
torch.cuda.synchronize(); gc.collect()
m0, r0, o0 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved(), len(gc.get_objects())
for _ in range(3000): step()
torch.cuda.synchronize(); gc.collect()
print('allocated', m0, '->', torch.cuda.memory_allocated(), 'reserved', r0, '->', torch.cuda.memory_reserved(), 'objects', o0, '->', len(gc.get_objects()), 'compiled calls', rz.compiled_calls)
