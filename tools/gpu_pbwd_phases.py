"""Developer probe (library built with -DEXA_PROBE_PBWD): phases of every wave of preprocess_bwd (100 MHz clock), C3 fwd + bwd.
The probe overwrites the z column of dL/dmean2D (always zero otherwise)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians
from exavatar_release_amd.camera import make_raster_matrices
dev = torch.device('cuda:0'); H = W = 1024; P = 150000
assets = scenes.dist_b_avatar(P, seed=0)
params = [assets[k].to(dev).requires_grad_(True) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
exa.config.mode = 'auto'
G = torch.randn(3, H, W, device=dev)
for k in (0, 50):
    tanx, tany, view, proj, cpos = make_raster_matrices(scenes.ring_camera(H, W, k, 200), (H, W))
    st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0, cpos.to(dev), False, False)
    m3, sc, rot, op, rgb = params
    for _ in range(3):
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        out = rasterize_gaussians(m3, m2, None, rgb, op, sc, rot, None, st)
        (out[0] * G).sum().backward()
    torch.cuda.synchronize()
    z = m2.grad[:, 2].cpu().numpy().reshape(-1)
    nw = P // 64
    z = z[: nw * 64].reshape(nw, 64)
    ph = z[:, :4] * 0.01            # us after the wave's start
    t0 = z[:, 4]; t0 = ((t0 - t0.min()) % (1 << 24)) * 0.01
    names = ('row 3 + heavy gather', 'own records gathered', 'chain rule', 'stores issued (end)')
    print('view %d: %d waves; wave starts: p50 %.2f p90 %.2f max %.2f us' % (k, nw, np.median(t0), np.percentile(t0, 90), t0.max()))
    for i, nm in enumerate(names):
        print('   after %-22s mean %5.2f p50 %5.2f p90 %5.2f max %5.2f us' % (nm, ph[:, i].mean(), np.median(ph[:, i]), np.percentile(ph[:, i], 90), ph[:, i].max()))
    print('   last end (start + end): %.2f us' % (t0 + ph[:, 3]).max())
