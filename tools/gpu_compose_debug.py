import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
dev = torch.device('cuda:0')
KEYS = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')
H, W, f = 128, 160, 170.0
scene = scenes.dist_a_random(3000, H, W, seed=51, focal=f)
human = scenes.dist_a_random(1500, H, W, seed=52, focal=f, z_range=(2.0, 4.0))
cam = {k: t.to(dev) for k, t in scenes.neutral_camera(H, W, focal=f).items()}
bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
G = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
rend = exa.GaussianRenderer()
exa.config.mode = 'exact'
res = {}
for how in ('merge', 'prefix', 'reference'):
    s = {k: v.to(dev).requires_grad_(True) for k, v in scene.items()}
    h = {k: v.to(dev).requires_grad_(True) for k, v in human.items()}
    r = {k: v.to(dev).requires_grad_(True) for k, v in human.items()}
    if how == 'reference':
        o = rend({k: torch.cat((s[k].detach(), h[k])) for k in KEYS}, (H, W), cam)
    else:
        out = exa.render_iteration(rend, s, h, r, (H, W), cam, bg, merge=(how == 'merge'))
        o = out['scene_human']
    (o['img'] * G).sum().backward()
    torch.cuda.synchronize()
    res[how] = ({k: h[k].grad.clone() for k in KEYS}, o['mean_2d'].grad.clone()[-1500:], o['img'].detach().clone())
print('img equal', torch.equal(res['merge'][2], res['prefix'][2]))
for pair in (('merge', 'reference'), ('prefix', 'reference')):
  print(pair)
  for k in KEYS:
    a, b = res[pair[0]][0][k], res[pair[1]][0][k]
    d = (a - b).abs().reshape(a.shape[0], -1).amax(1)
    print(k, 'max diff %.3g of %.3g; rows differing %d / %d; merge all-zero rows %d, prefix all-zero rows %d' % (
        float(d.max()), float(b.abs().max()), int((d > 1e-5 * float(b.abs().max())).sum()), a.shape[0],
        int((a.reshape(a.shape[0], -1).abs().amax(1) == 0).sum()), int((b.reshape(b.shape[0], -1).abs().amax(1) == 0).sum())))
a, b = res['merge'][1], res['prefix'][1]
print('mean_2d max diff', float((a - b).abs().max()), float(b.abs().max()))
bad = ((res['merge'][0]['rgb'] - res['prefix'][0]['rgb']).abs().amax(1) > 1e-4).nonzero().flatten()
print('first bad rows', bad[:20].tolist())
for i in bad[:5].tolist():
    print(i, res['merge'][0]['rgb'][i].tolist(), res['prefix'][0]['rgb'][i].tolist())
