import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from oracle import raster_oracle as ro
dev = torch.device('cuda:0')
H, W = 135, 240
f = 1.2 * max(H, W)
assets = scenes.dist_a_random(2500, H, W, seed=H * 7 + W, focal=f)
cam = scenes.neutral_camera(H, W, focal=f)
g = torch.Generator().manual_seed(3)
bg = torch.rand(3, generator=g)
with torch.no_grad():
    out = exa.GaussianRenderer()({k: v.to(dev) for k, v in assets.items()}, (H, W), {k: v.to(dev) for k, v in cam.items()}, bg.to(dev))
    ref = ro.render(assets, (H, W), cam, bg, return_aux=True)
d = (out['img'].cpu() - ref['img']).abs().amax(0)
amb = ro.ambiguous_pixel_mask(ref['aux'], H, W)
d[amb] = 0
print('bad pixels', int((d > 1e-4).sum()))
ys, xs = torch.nonzero(d > 1e-4, as_tuple=True)
aux = ref['aux']; pre = aux['pre']
for y, x in list(zip(ys.tolist(), xs.tolist()))[:6]:
    print('pixel', y, x, 'diff', float(d[y, x]), 'hip', out['img'][:, y, x].tolist(), 'ref', ref['img'][:, y, x].tolist())
    t = (y // 16) * pre['grid'][0] + (x // 16)
    s0, e0 = aux['ranges'][t].tolist()
    ids = aux['sorted_idx'][s0:e0]
    T = 1.0
    for gi in ids.tolist():
        dx = float(pre['px'][gi]) - x; dy = float(pre['py'][gi]) - y
        A, B, C = [float(v) for v in pre['conic'][gi]]
        p = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
        o = float(assets['opacity'][gi])
        al = min(0.99, o * math.exp(p)) if p <= 0 else 0
        if al >= 1 / 255.:
            a2, b2, c2 = [float(v[gi]) for v in pre['cov2']]
            tau2 = 2 * math.log(255 * o)
            ex, ey = math.sqrt(tau2 * a2), math.sqrt(tau2 * c2)
            inside = abs(dx) <= ex and abs(dy) <= ey
            print('   g=%d alpha=%.5f T=%.4f contrib=%.5f px=%.3f py=%.3f dx=%.3f dy=%.3f ex=%.3f ey=%.3f inside_bbox=%s radius=%d rect=%s o=%.4f cov=(%.3f %.3f %.3f)' % (
                gi, al, T, al * T, float(pre['px'][gi]), float(pre['py'][gi]), dx, dy, ex, ey, inside, int(pre['radius'][gi]),
                [int(v[gi]) for v in pre['rect']], o, a2, b2, c2))
            T *= (1 - al)
print('pixel_margin[8,239] =', float(aux['pixel_margin'][8, 239]), 'amb =', bool(amb[8, 239]), 'threads', torch.get_num_threads())
print('min margin overall', float(aux['pixel_margin'].min()), 'n amb', int(amb.sum()))
