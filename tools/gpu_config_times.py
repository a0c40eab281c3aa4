"""Developer tool: per-kernel HIP-event times and instance statistics for the other BASELINE configs (scene + avatar mixes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import exavatar_release_amd as exa
from exavatar_release_amd import scenes, _lib
from exavatar_release_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians, last_header, _debug_last
from exavatar_release_amd.camera import make_raster_matrices

dev = torch.device('cuda:0')
exa.config.mode = 'exact'
exa.config.keep_debug = True
for name in (sys.argv[1:] or ['c2', 'c3s', 'c5']):
    assets, (H, W), cam = scenes.make_config(name)
    P = assets['mean_3d'].shape[0]
    params = [assets[k].to(dev).requires_grad_(True) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
    mean_2d = torch.zeros(P, 3, device=dev, requires_grad=True)
    G = torch.randn(3, H, W, device=dev)
    tanx, tany, view, proj, cpos = make_raster_matrices(cam, (H, W))
    st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0, cpos.to(dev), False, False)
    acc = {}
    for rep in range(4):
        _lib.timing_enable(rep > 0)
        m3, sc, rot, op, rgb = params
        color, radii, depth, alpha = rasterize_gaussians(m3, mean_2d, None, rgb, op, sc, rot, None, st)
        torch.autograd.grad([color], params + [mean_2d], grad_outputs=[G])
        torch.cuda.synchronize()
        if rep > 0:
            for n, v in _lib.timing_read().items():
                acc[n] = acc.get(n, 0.0) + v / 3 * 1e3
    _lib.timing_enable(False)
    hdr = last_header()
    n = _debug_last['geom'].view(torch.int32).view(-1, 16)[:P, 14].cpu().numpy()
    print('%s: P=%d %dx%d visible=%d instances=%d (n_inst mean %.1f p99 %d max %d)' % (name, P, H, W, hdr[3], hdr[4], n[n > 0].mean(), np.percentile(n[n > 0], 99), n.max()))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _layout import tile_offsets
    lay = tile_offsets(P, W, H)
    r = _debug_last['tile'][lay['ranges'][0]: lay['ranges'][0] + lay['ranges'][1]].view(torch.int32).view(-1, 2).cpu().numpy().astype(np.int64)
    ln = r[:, 1] - r[:, 0]; ln = ln[ln > 0]
    print('   lists: %d non-empty, mean %.0f p50 %d p99 %d max %d; > 1024: %d, > 2048: %d, > 4096: %d' % (len(ln), ln.mean(), np.median(ln), np.percentile(ln, 99), ln.max(), (ln > 1024).sum(), (ln > 2048).sum(), (ln > 4096).sum()))
    print('   ' + '  '.join('%s=%.1f' % (k, v) for k, v in acc.items()) + '  total=%.1f us' % sum(acc.values()))
