"""Developer probe (library built with -DEXA_PROBE_PFWD): phases of every preprocess_fwd workgroup (100 MHz clock), C3.  The probe
overwrites the radii of the first Gaussians of every chunk."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians
from exavatar_release_amd.camera import make_raster_matrices
dev = torch.device('cuda:0'); H = W = 1024; P = 150000
assets = scenes.dist_b_avatar(P, seed=0)
params = [assets[k].to(dev) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
exa.config.mode = 'auto'
for k in (0, 50):
    tanx, tany, view, proj, cpos = make_raster_matrices(scenes.ring_camera(H, W, k, 200), (H, W))
    st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0, cpos.to(dev), False, False)
    m3, sc, rot, op, rgb = params
    with torch.no_grad():
        for _ in range(4):
            out = rasterize_gaussians(m3, torch.zeros(P, 3, device=dev), None, rgb, op, sc, rot, None, st)
    torch.cuda.synchronize()
    radii = out[1].cpu().numpy().astype(np.int64)
    nchunk = (P + 1023) // 1024
    r = np.stack([radii[c * 1024: c * 1024 + 4] for c in range(nchunk)])
    ph = r[:, 1:3] * 0.01
    t0 = (r[:, 3] - r[:, 3].min()) % (1 << 24) * 0.01
    print('view %d: %d workgroups; starts within %.2f us' % (k, nchunk, t0.max()))
    for i, nm in enumerate(('projected + records + histogram', 'totals + matrix row (end)')):
        print('   after %-32s mean %5.2f p90 %5.2f max %5.2f us' % (nm, ph[:, i].mean(), np.percentile(ph[:, i], 90), ph[:, i].max()))
    print('   last end (start + end): %.2f us' % (t0 + ph[:, 1]).max())
