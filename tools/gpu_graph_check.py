"""Developer check: hipGraph replay of fwd+bwd must equal eager results."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faulthandler; faulthandler.enable()
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians
from exavatar_release_amd.camera import make_raster_matrices

dev = torch.device('cuda:0')
H = W = 512
assets = scenes.dist_b_avatar(20000, seed=0)
params = [assets[k].to(dev).requires_grad_(True) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
P = params[0].shape[0]
mats = [make_raster_matrices(scenes.ring_camera(H, W, k, 8), (H, W)) for k in range(8)]
view_s = mats[0][2].to(dev).clone(); proj_s = mats[0][3].to(dev).clone(); cpos_s = mats[0][4].to(dev).clone()
st = GaussianRasterizationSettings(H, W, mats[0][0], mats[0][1], torch.ones(3, device=dev), 1.0, view_s, proj_s, 0, cpos_s, False, False)
mean_2d = torch.zeros(P, 3, device=dev, requires_grad=True)
G = torch.randn(3, H, W, device=dev)
outs = {}

def step():
    m3, sc, rot, op, rgb = params
    color, radii, depth, alpha = rasterize_gaussians(m3, mean_2d, None, rgb, op, sc, rot, None, st)
    grads = torch.autograd.grad([color], params + [mean_2d], grad_outputs=[G])
    outs['color'] = color; outs['grads'] = grads; outs['radii'] = radii
    return color, grads

def set_view(i):
    view_s.copy_(mats[i][2].to(dev)); proj_s.copy_(mats[i][3].to(dev)); cpos_s.copy_(mats[i][4].to(dev))

exa.config.mode = 'exact'
ref = []
for i in range(3):
    set_view(i)
    c, g = step()
    ref.append((c.clone(), [x.clone() for x in g]))
torch.cuda.synchronize()
exa.config.mode = 'capacity'; exa.config.fixed_capacity = 400000
set_view(0)
for _ in range(2): step()
torch.cuda.synchronize(); exa.check_overflow()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize(); exa.check_overflow()
print('capturing', flush=True)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    step()
print('captured', flush=True)
for i in range(3):
    set_view(i)
    gr.replay()
    torch.cuda.synchronize()
    c, g = outs['color'], outs['grads']
    from exavatar_release_amd.rasterizer import last_header
    print('header', last_header())
    print('view', i, 'img diff', float((c - ref[i][0]).abs().max()), 'vis', int((outs['radii'] > 0).sum()),
          'grad diffs', [float((a - b).abs().max() / b.abs().max().clamp_min(1e-20)) for a, b in zip(g, ref[i][1])])
