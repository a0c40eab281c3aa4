"""Developer tool: histogram of sub-tile list lengths for a few ring views of config C3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians, _debug_last
from exavatar_release_amd.camera import make_raster_matrices
dev = torch.device('cuda:0'); H = W = 1024; P = 150000
assets = scenes.dist_b_avatar(P, seed=0)
params = [assets[k].to(dev) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
exa.config.mode = 'exact'
cells = 256; from _layout import tile_offsets
lay = tile_offsets(P, W, H); cells = lay['cells']; off = lay['slots'][0]
for k in [int(v) for v in (sys.argv[1:] or [0, 25, 50])]:
    tanx, tany, view, proj, cpos = make_raster_matrices(scenes.ring_camera(H, W, k, 200), (H, W))
    st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0, cpos.to(dev), False, False)
    m3, sc, rot, op, rgb = params
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    rasterize_gaussians(m3, m2, None, rgb, op, sc, rot, None, st)
    torch.cuda.synchronize()
    tile = _debug_last['tile']
    slots = tile[off: off + cells * 64 * 16].view(torch.int32).view(-1, 4).cpu().numpy().astype(np.int64)
    n = slots[:, 1] - slots[:, 0]
    act = n[n > 0]
    print('view %d: active %d sum %d mean %.0f p50 %d p90 %d p99 %d max %d ; >512: %d  >1024: %d  >2048: %d ; first 8 in launch order: %s' % (
        k, len(act), act.sum(), act.mean(), np.median(act), np.percentile(act, 90), np.percentile(act, 99), act.max(),
        (act > 512).sum(), (act > 1024).sum(), (act > 2048).sum(), n[:8].tolist()))
