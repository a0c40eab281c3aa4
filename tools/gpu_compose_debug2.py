import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes, renderer as rn, rasterizer as rz
dev = torch.device('cuda:0')
KEYS = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')
H, W, f = 128, 160, 170.0
cam = {k: t.to(dev) for k, t in scenes.neutral_camera(H, W, focal=f).items()}
G = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
rend = exa.GaussianRenderer()
exa.config.mode = 'exact'


def cmp(tag, a, b):
    for k in KEYS:
        d = (a[k] - b[k]).abs().reshape(a[k].shape[0], -1).amax(1)
        print(tag, k, 'max diff %.3g of %.3g; rows differing %d / %d' % (float(d.max()), float(b[k].abs().max()),
              int((d > 1e-5 * float(b[k].abs().max())).sum()), a[k].shape[0]))


def composite_only(scene, human, K1=True):
    s = {k: v.to(dev).requires_grad_(True) for k, v in scene.items()}
    h = {k: v.to(dev).requires_grad_(True) for k, v in human.items()}
    plain = [rn._raster_job(s, (H, W), cam, None), rn._raster_job(h, (H, W), cam, None)]
    outs, handles = rz.rasterize_gaussians_batch(plain, keep_keys=True)
    comp = [rn._raster_job(h, (H, W), cam, None)]
    co = rz.rasterize_composites([(handles[0], handles[1])], comp)[0]
    (co[0] * G).sum().backward()
    torch.cuda.synchronize()
    return {k: h[k].grad.clone() for k in KEYS}, co[0].detach().clone(), handles


def reference(scene, human):
    s = {k: v.to(dev) for k, v in scene.items()}
    h = {k: v.to(dev).requires_grad_(True) for k, v in human.items()}
    o = rend({k: torch.cat((s[k], h[k])) for k in KEYS}, (H, W), cam)
    (o['img'] * G).sum().backward()
    torch.cuda.synchronize()
    return {k: h[k].grad.clone() for k in KEYS}, o['img'].detach().clone()


scene = scenes.dist_a_random(3000, H, W, seed=51, focal=f)
human = scenes.dist_a_random(1500, H, W, seed=52, focal=f, z_range=(2.0, 4.0))
for name, sc in (('scene as is', scene), ('scene opacity 0', dict(scene, opacity=torch.zeros_like(scene['opacity']))),
                 ('scene far behind', dict(scene, mean_3d=scene['mean_3d'] + torch.tensor([0.0, 0.0, 50.0])))):
    g1, i1, hd = composite_only(sc, human)
    g2, i2 = reference(sc, human)
    print('==', name, 'img equal', torch.equal(i1, i2))
    cmp(name, g1, g2)
# tiny human: one Gaussian
h1 = {k: v[:1].clone() for k, v in human.items()}
g1, i1, hd = composite_only(scene, h1)
g2, i2 = reference(scene, h1)
print('== one human Gaussian: img equal', torch.equal(i1, i2))
for k in KEYS:
    print(k, g1[k].flatten().tolist(), g2[k].flatten().tolist())
jb = hd[1]
print('B capacity', jb.capacity, 'A capacity', hd[0].capacity)
print('---- pattern')
for n in (2, 5, 64, 65, 300):
    hn = {k: v[:n].clone() for k, v in human.items()}
    g1, i1, hd = composite_only(scene, hn)
    g2, i2 = reference(scene, hn)
    d = (g1['rgb'] - g2['rgb']).abs().amax(1)
    sc = float(g2['rgb'].abs().max())
    badrows = (d > 1e-5 * sc).nonzero().flatten().tolist()
    zero_bad = [i for i in badrows if float(g1['rgb'][i].abs().max()) == 0.0]
    print('n', n, 'img equal', torch.equal(i1, i2), 'bad rows', len(badrows), badrows[:16], 'of which all-zero', len(zero_bad))
print('---- n = 2 detail')
hn = {k: v[:2].clone() for k, v in human.items()}
g1, i1, hd = composite_only(scene, hn)
g2, i2 = reference(scene, hn)
for k in ('rgb', 'opacity', 'mean_3d'):
    print(k, 'composite', g1[k].tolist(), 'reference', g2[k].tolist())
jb = hd[1]
sp = jb.ws[:2 * 64].view(torch.int32).view(2, 16).cpu()
print('B splat rows 3:', sp[:, 12:16].tolist(), 'radii', jb.radii.tolist())
# swap the order of the two human Gaussians
hs = {k: v[:2].flip(0).clone() for k, v in human.items()}
g3, i3, _ = composite_only(scene, hs)
print('swapped rgb', g3['rgb'].tolist())
