import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, numpy as np
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians, _debug_last
from exavatar_release_amd.camera import make_raster_matrices
dev = torch.device('cuda:0'); H = W = 1024; P = 150000
assets = scenes.dist_b_avatar(P, seed=0)
params = [assets[k].to(dev) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
exa.config.mode = 'exact'
for k in (0, 50):
    tanx, tany, view, proj, cpos = make_raster_matrices(scenes.ring_camera(H, W, k, 200), (H, W))
    st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0, cpos.to(dev), False, False)
    m3, sc, rot, op, rgb = params
    rasterize_gaussians(m3, torch.zeros(P, 3, device=dev, requires_grad=True), None, rgb, op, sc, rot, None, st)
    torch.cuda.synchronize()
    g = _debug_last['geom'].view(torch.int32).view(-1, 16)[:P].cpu().numpy()
    n = g[:, 14]
    print('view', k, 'n_inst: mean %.1f p50 %d p90 %d p99 %d p99.9 %d max %d; >16: %d >32: %d >64: %d' % (n.mean(), np.median(n), np.percentile(n, 90), np.percentile(n, 99), np.percentile(n, 99.9), n.max(), (n > 16).sum(), (n > 32).sum(), (n > 64).sum()))
    wm = n.reshape(-1, 64)[: P // 64].max(axis=1) if P % 64 == 0 else n[: P // 64 * 64].reshape(-1, 64).max(axis=1)
    print('   per-wave max n_inst: mean %.1f p50 %d p90 %d max %d' % (wm.mean(), np.median(wm), np.percentile(wm, 90), wm.max()))
