"""Developer probe (library built with EXA_PROBE_SORT=1): cycles of every sort workgroup versus list length."""
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
exa.config.mode = 'exact'
lay = tile_offsets(P, W, H)
for k in (0, 50):
    tanx, tany, view, proj, cpos = make_raster_matrices(scenes.ring_camera(H, W, k, 200), (H, W))
    st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0, cpos.to(dev), False, False)
    m3, sc, rot, op, rgb = params
    with torch.no_grad():
        for _ in range(2):
            rasterize_gaussians(m3, torch.zeros(P, 3, device=dev), None, rgb, op, sc, rot, None, st)
    torch.cuda.synchronize()
    tile = _debug_last['tile']
    r = tile[lay['ranges'][0]: lay['ranges'][0] + lay['ranges'][1]].view(torch.int32).view(-1, 2).cpu().numpy().astype(np.int64)
    cyc = tile[lay['part_cnt'][0]: lay['part_cnt'][0] + lay['cells'] * 64 * 4].view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    n = r[:, 1] - r[:, 0]
    print('view', k)
    for lo, hi in ((1, 64), (65, 128), (129, 256), (257, 512), (513, 1024), (1025, 2048)):
        m = (n >= lo) & (n <= hi)
        if m.any():
            print('   n in [%d,%d]: %5d lists, cycles mean %7.0f p90 %7.0f max %7.0f  (%.0f cycles per key)' % (lo, hi, m.sum(), cyc[m].mean(), np.percentile(cyc[m], 90), cyc[m].max(), (cyc[m] / n[m]).mean()))
    print('   total WG-cycles %.3g ; if spread over 256 CUs x 5 WG: %.1f us at 2.1 GHz' % (cyc[n > 0].sum(), cyc[n > 0].sum() / (256 * 5) / 2100))
