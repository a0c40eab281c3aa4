"""Developer probe (library built with -DEXA_PROBE_SCATTER): phases of every cell_scatter workgroup (100 MHz clock), C3 fused call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, numpy as np
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians, _debug_last
from exavatar_release_amd.camera import make_raster_matrices
from _layout import tile_offsets
dev = torch.device('cuda:0'); H = W = 1024; P = 150000
assets = scenes.dist_b_avatar(P, seed=0)
params = [assets[k].to(dev) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
exa.config.mode = 'auto'; exa.config.keep_debug = True
lay = tile_offsets(P, W, H); nwg = lay['chunks'] + 1
for k in (0, 50):
    tanx, tany, view, proj, cpos = make_raster_matrices(scenes.ring_camera(H, W, k, 200), (H, W))
    st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0, cpos.to(dev), False, False)
    m3, sc, rot, op, rgb = params
    for _ in range(4):
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        rasterize_gaussians(m3, m2, None, rgb, op, sc, rot, None, st)
    torch.cuda.synchronize()
    tile = _debug_last['tile']
    pc = tile[lay['part_cnt'][0]: lay['part_cnt'][0] + lay['part_cnt'][1]].view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    r = pc[40000: 40000 + 8 * nwg].reshape(nwg, 8)
    ph = r[:, :6] * 0.01; t0 = r[:, 6] * 0.01; t0 = t0 - t0.min()
    sc_ = slice(0, nwg - 1)
    print('view %d: %d scatter workgroups + publisher; starts within %.2f us' % (k, nwg - 1, t0.max()))
    for i, name in enumerate(('zero-fill issued', 'column walk', 'block scans', 'splat rows + prefix', 'entries scattered (end)')):
        print('   after %-26s mean %5.2f  p90 %5.2f  max %5.2f us' % (name, ph[sc_, i].mean(), np.percentile(ph[sc_, i], 90), ph[sc_, i].max()))
    print('   publisher: column walk %.2f, end %.2f us (start %.2f)' % (ph[nwg - 1, 1], ph[nwg - 1, 5], t0[nwg - 1]))
    print('   last end over all (start + end): %.2f us' % max((t0[sc_] + ph[sc_, 4]).max(), t0[nwg - 1] + ph[nwg - 1, 5]))
