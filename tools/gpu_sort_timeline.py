"""Developer probe (library built with -DEXA_PROBE_SORTLINE): start / end of every workgroup of sort_subtiles_kernel (100 MHz
clock) against the length of its list; the ordering workgroups come first.  C3, forward only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, numpy as np
import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.rasterizer import GaussianRasterizationSettings, rasterize_gaussians, _debug_last
from exavatar_release_amd.camera import make_raster_matrices
from _layout import tile_offsets
OW = 32          # ORDER_WGS of csrc/render_fwd.hip
dev = torch.device('cuda:0'); H = W = 1024; P = 150000
assets = scenes.dist_b_avatar(P, seed=0)
params = [assets[k].to(dev) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
exa.config.mode = 'exact'; exa.config.keep_debug = True
lay = tile_offsets(P, W, H); cells = lay['cells']; nsub = cells * 64
for k in [int(v) for v in (sys.argv[1:] or [0, 50])]:
    tanx, tany, view, proj, cpos = make_raster_matrices(scenes.ring_camera(H, W, k, 200), (H, W))
    st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0, cpos.to(dev), False, False)
    m3, sc, rot, op, rgb = params
    with torch.no_grad():
        for _ in range(3):
            rasterize_gaussians(m3, torch.zeros(P, 3, device=dev), None, rgb, op, sc, rot, None, st)
    torch.cuda.synchronize()
    tile = _debug_last['tile']
    desc = tile[lay['cell_desc'][0]: lay['cell_desc'][0] + lay['cell_desc'][1]].view(torch.int32).view(-1, 4).cpu().numpy().astype(np.int64)
    rng = tile[lay['ranges'][0]: lay['ranges'][0] + nsub * 8].view(torch.int32).view(-1, 2).cpu().numpy().astype(np.int64)
    tt = tile[lay['part_cnt'][0]: lay['part_cnt'][0] + (nsub + OW) * 8].view(torch.int32).view(-1, 2).cpu().numpy().astype(np.int64) & 0xffffffff
    start, end = tt[:, 0] * 0.01, tt[:, 1] * 0.01
    # sorting workgroup OW + w handles sub-tile (w & 63) of the cell of rank (w >> 6)
    wg = np.arange(nsub)
    stile = desc[wg >> 6, 0] * 64 + (wg & 63)
    n = rng[stile, 1] - rng[stile, 0]
    t0 = min(start[:OW].min(), start[OW:][n > 0].min())
    start -= t0; end -= t0
    so, eo = start[:OW], end[:OW]
    print('view %d: ordering workgroups start %.2f..%.2f end %.2f..%.2f us' % (k, so.min(), so.max(), eo.min(), eo.max()))
    ph = tile[lay['part_cnt'][0] + (nsub + OW) * 8: lay['part_cnt'][0] + (nsub + OW) * 8 + OW * 8 * 4].view(torch.int32).view(OW, 8).cpu().numpy().astype(np.int64) & 0xffffffff
    ph = ph[:, :6] * 0.01 - t0
    print('   phases of the ordering workgroups (mean end, us after the first start): entry %.2f | histogram %.2f | prefixes %.2f | ranks %.2f | '
          'reservations %.2f | records %.2f' % tuple(ph.mean(0)))
    s_, e_ = start[OW:], end[OW:]
    work = n > 0
    d = e_ - s_
    print('   %d sorting workgroups with a list: starts p50 %.2f p90 %.2f max %.2f; ends p50 %.2f p90 %.2f p99 %.2f max %.2f us' % (
        work.sum(), *[np.percentile(s_[work], q) for q in (50, 90, 100)], *[np.percentile(e_[work], q) for q in (50, 90, 99, 100)]))
    for lo, hi in ((1, 64), (65, 256), (257, 512), (513, 1024), (1025, 2048), (2049, 1 << 20)):
        m = work & (n >= lo) & (n <= hi)
        if m.any():
            print('   n in [%4d,%5d]: %5d lists, duration mean %5.2f max %5.2f us, start mean %5.2f' % (lo, min(hi, n.max()), m.sum(), d[m].mean(), d[m].max(), s_[m].mean()))
    last = np.argsort(-(e_ * work))[:5]
    print('   last to end: ' + '; '.join('wg %d n %d start %.2f dur %.2f' % (i, n[i], s_[i], d[i]) for i in last))
    idle = ~work
    print('   workgroups without a list: %d, last start %.2f us' % (idle.sum(), s_[idle & (e_ > 0)].max() if (idle & (e_ > 0)).any() else -1))
