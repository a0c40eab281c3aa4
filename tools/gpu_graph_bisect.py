"""Developer tool: which ingredient makes hipGraph capture of fwd+bwd crash?  Each variant runs in a subprocess."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = r'''
import sys, faulthandler; faulthandler.enable()
sys.path.insert(0, %r)
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.camera import make_raster_matrices
v = sys.argv[1].split(',')
dev = torch.device('cuda:0')
H = W = 384
KEYS = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')
assets = scenes.dist_b_avatar(30000, seed=2)
params = [assets[k].to(dev).requires_grad_(True) for k in KEYS]
m = make_raster_matrices(scenes.ring_camera(H, W, 0, 12, focal=560.0), (H, W))
st = exa.GaussianRasterizationSettings(H, W, m[0], m[1], torch.ones(3, device=dev), 1.0, m[2].to(dev), m[3].to(dev), 0, m[4].to(dev), False, False)
m2 = torch.zeros(30000, 3, device=dev, requires_grad=True)
G = torch.randn(3, H, W, device=dev)
holder = {}
def step():
    m3, sc, rot, op, rgb = params
    col, rad, dep, alp = exa.rasterize_gaussians(m3, m2, None, rgb, op, sc, rot, None, st)
    ins = params + [m2] if 'm2' in v else params
    if 'fwdonly' in v:
        out = col
    else:
        out = torch.autograd.grad([col], ins, grad_outputs=[G])
    if 'hold' in v:
        holder['col'] = col; holder['g'] = out
if 'exactfirst' in v:
    exa.config.mode = 'exact'
    for i in range(2): step()
exa.config.mode = 'capacity'; exa.config.fixed_capacity = 2000000
if 'warm3' in v:
    for i in range(3): step()
    torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for i in range(2 if 'side2' in v else 1): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
g.replay(); torch.cuda.synchronize()
print('OK', sys.argv[1])
''' % ROOT
variants = ['base', 'hold', 'm2', 'warm3', 'warm3,side2,m2', 'exactfirst', 'exactfirst,hold', 'fwdonly', 'fwdonly,hold',
            'exactfirst,warm3', 'hold,warm3']
for var in (sys.argv[1:] or variants):
    r = subprocess.run([sys.executable, '-c', VARIANT, var], capture_output=True, text=True, timeout=120)
    tail = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith('OK') or 'Fatal' in l or 'Error' in l][:2]
    print('%-22s rc=%d %s' % (var, r.returncode, tail))
