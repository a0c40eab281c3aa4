"""Developer check (run through gpurun): HIP path vs the CPU oracle on a named config, verbose."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exavatar_release_amd import GaussianRenderer, scenes
from oracle import raster_oracle as ro


def run(name, do_oracle=True):
    assets, shp, cam = scenes.make_config(name)
    H, W = shp
    g = torch.Generator().manual_seed(1)
    G = torch.randn(3, H, W, generator=g)
    bg = torch.rand(3, generator=g)
    dev = torch.device('cuda:0')
    a_gpu = {k: v.to(dev).requires_grad_(True) for k, v in assets.items()}
    cam_gpu = {k: v.to(dev) for k, v in cam.items()}
    rend = GaussianRenderer()
    torch.cuda.synchronize()
    t = time.time()
    out = rend(a_gpu, shp, cam_gpu, bg.to(dev))
    torch.cuda.synchronize()
    print(name, 'hip fwd (first call) %.3f s' % (time.time() - t))
    (out['img'] * G.to(dev)).sum().backward()
    torch.cuda.synchronize()
    print(name, 'hip fwd+bwd done; vis', int(out['is_vis'].sum()), 'img range', float(out['img'].min()), float(out['img'].max()))
    if not do_oracle:
        return
    a_cpu = {k: v.clone().requires_grad_(True) for k, v in assets.items()}
    t = time.time()
    ref = ro.render(a_cpu, shp, cam, bg, return_aux=True)
    print(name, 'oracle fwd %.2f s' % (time.time() - t))
    t = time.time()
    (ref['img'] * G).sum().backward()
    print(name, 'oracle bwd %.2f s' % (time.time() - t))
    amb = ro.ambiguous_pixel_mask(ref['aux'], H, W, include_gaussians=False)
    print('ambiguous pixels', int(amb.sum()))
    for k, rk in (('img', 'img'), ('depthmap', 'depthmap'), ('mask', 'mask')):
        d = (out[k].detach().cpu() - ref[rk].detach()).abs()
        dm = d.clone()
        dm[..., amb] = 0
        print('%-9s Linf all %.3e   Linf non-ambiguous %.3e   #>1e-4: %d' % (k, d.max(), dm.max(), int((dm > 1e-4).sum())))
    print('radii equal:', bool((out['radius'].cpu() == ref['radius']).all()), 'mismatches', int((out['radius'].cpu() != ref['radius']).sum()))
    for k in assets:
        gh = a_gpu[k].grad.cpu()
        gr = a_cpu[k].grad
        print('grad %-9s max|ref| %.3e  max|diff| %.3e  relL2 %.3e' % (k, gr.abs().max(), (gh - gr).abs().max(), (gh - gr).norm() / gr.norm().clamp_min(1e-30)))
    gh = out['mean_2d'].grad.cpu(); gr = ref['mean_2d'].grad
    print('grad %-9s max|ref| %.3e  max|diff| %.3e  relL2 %.3e' % ('mean_2d', gr.abs().max(), (gh - gr).abs().max(), (gh - gr).norm() / gr.norm().clamp_min(1e-30)))


if __name__ == '__main__':
    for n in sys.argv[1:] or ['c1']:
        run(n)
