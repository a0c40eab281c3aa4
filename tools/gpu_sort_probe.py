"""Developer probe (needs a library built with -DEXA_PROBE_SORT): per-sub-tile cycles of the sort kernel."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import exavatar_release_amd as exa
from exavatar_release_amd import scenes, _lib
from exavatar_release_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians, _debug_last
from exavatar_release_amd.camera import make_raster_matrices
dev = torch.device('cuda:0'); H = W = 1024; P = 150000
assets = scenes.dist_b_avatar(P, seed=0)
params = [assets[k].to(dev) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
tanx, tany, view, proj, cpos = make_raster_matrices(scenes.ring_camera(H, W, k, 200), (H, W))
st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0, cpos.to(dev), False, False)
exa.config.mode = 'exact'
for rep in range(3):
    if True:
        m3, sc, rot, op, rgb = params
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        rasterize_gaussians(m3, m2, None, rgb, op, sc, rot, None, st)
    torch.cuda.synchronize()
tile = _debug_last['tile']
from _layout import tile_offsets
lay = tile_offsets(P, W, H); cells = lay['cells']; off = lay['slots'][0]
slots = tile[off: off + cells * 64 * 16].view(torch.int32).view(-1, 4).cpu().numpy().astype(np.int64)
n = slots[:, 1] - slots[:, 0]; cyc = slots[:, 3] & 0xffffffff
act = n > 0
print('active', act.sum(), 'cycles: mean %.0f  p50 %.0f  p99 %.0f  max %.0f' % (cyc[act].mean(), np.median(cyc[act]), np.percentile(cyc[act], 99), cyc[act].max()))
for lo, hi in ((1, 64), (65, 128), (129, 256), (257, 512), (513, 1024), (1025, 100000)):
    m = (n >= lo) & (n <= hi)
    if m.any(): print('n in [%d,%d]: count %d  mean cycles %.0f  max %.0f' % (lo, hi, m.sum(), cyc[m].mean(), cyc[m].max()))
hw = slots[:, 2] & 0xffffffff
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = (hw >> 16) & 15
key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
ka = key[act]
uniq, inv = np.unique(ka, return_inverse=True)
work = np.bincount(inv, weights=cyc[act].astype(np.float64) * 0 + n[act]); cnt = np.bincount(inv)
print('SIMDs with active waves: %d ; active waves per SIMD: mean %.2f max %d ; sum n per SIMD: mean %.0f p90 %.0f max %.0f' % (len(uniq), cnt.mean(), cnt.max(), work.mean(), np.percentile(work, 90), work.max()))
cuk = ka // 4; u2, inv2 = np.unique(cuk, return_inverse=True); w2 = np.bincount(inv2, weights=n[act].astype(np.float64))
print('CUs with active waves: %d ; sum n per CU: mean %.0f p90 %.0f max %.0f min %.0f' % (len(u2), w2.mean(), np.percentile(w2, 90), w2.max(), w2.min()))
xk = xcc[act]; print('per XCC sum n:', [int(n[act][xk == x].sum()) for x in range(8)])
endc = np.zeros(len(uniq)); np.maximum.at(endc, inv, cyc[act]); print('per-SIMD last wave end cycles: mean %.0f p90 %.0f max %.0f' % (endc.mean(), np.percentile(endc, 90), endc.max()))
idx = np.argsort(-work)[:5]; print('heaviest SIMDs (sum n, waves, end):', [(int(work[i]), int(cnt[i]), int(endc[i])) for i in idx])
print('first 24 launch idx -> (xcc,se,sh,cu,simd):', [(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(simd[i])) for i in np.nonzero(act)[0][:24]])
order = np.argsort(-cyc)[:5]
print('slowest slots (launch idx, n, cycles):', [(int(i), int(n[i]), int(cyc[i])) for i in order])
