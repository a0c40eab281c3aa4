"""Developer probe (library built with -DEXA_PROBE_FWD): when every wave of render_fwd starts and ends (chip-wide 100 MHz
clock), against the length of its list -- is the launch as long as its longest list?  C3, two ring views, training forward."""
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
exa.config.mode = 'exact'; exa.config.keep_debug = True
lay = tile_offsets(P, W, H); cells = lay['cells']; nsub = cells * 64
for k in [int(v) for v in (sys.argv[1:] or [0, 50, 123])]:
    tanx, tany, view, proj, cpos = make_raster_matrices(scenes.ring_camera(H, W, k, 200), (H, W))
    st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0, cpos.to(dev), False, False)
    m3, sc, rot, op, rgb = params
    for _ in range(3):
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        rasterize_gaussians(m3, m2, None, rgb, op, sc, rot, None, st)
    torch.cuda.synchronize()
    tile = _debug_last['tile']
    slots = tile[lay['slots'][0]: lay['slots'][0] + nsub * 16].view(torch.int32).view(-1, 4).cpu().numpy().astype(np.int64)
    tt = tile[lay['part_cnt'][0]: lay['part_cnt'][0] + nsub * 16].view(torch.int32).view(-1, 4).cpu().numpy().astype(np.int64) & 0xffffffff
    ex = tile[lay['fwd_exit'][0]: lay['fwd_exit'][0] + nsub * 8].view(torch.int32).view(-1, 2).cpu().numpy().astype(np.int64)
    n = slots[:, 1] - slots[:, 0]
    entered = ex[slots[:, 2], 1]
    work = n > 0
    start, end = tt[:, 0] * 0.01, tt[:, 1] * 0.01        # us
    t0 = start[work].min()
    start -= t0; end -= t0
    dur = end - start
    print('view %d: %d non-empty lists, %d entries, longest %d; span of the launch (first start -> last end, non-empty) %.2f us' % (k, work.sum(), n.sum(), n.max(), end[work].max()))
    print('   waves done at: 50 %% %.1f us, 90 %% %.1f, 99 %% %.1f, all %.1f ; starts: 50 %% %.1f, 90 %% %.1f, last %.1f' % (*[np.percentile(end[work], q) for q in (50, 90, 99, 100)], *[np.percentile(start[work], q) for q in (50, 90, 100)]))
    order = np.argsort(-end * work)[:6]
    print('   last to end: ' + '; '.join('slot %d n %d batches %d start %.1f dur %.1f' % (i, n[i], entered[i], start[i], dur[i]) for i in order))
    for lo, hi in ((1, 64), (65, 128), (129, 256), (257, 512), (513, 1024), (1025, 4096)):
        m = work & (n >= lo) & (n <= hi)
        if m.any():
            walked = np.minimum(n[m], entered[m] * 64)
            print('   n in [%4d,%4d]: %5d lists, duration mean %5.1f max %5.1f us, %.0f ns per walked entry' % (lo, hi, m.sum(), dur[m].mean(), dur[m].max(), (dur[m] * 1e3 / np.maximum(walked, 1)).mean()))
    hw = tt[:, 2]
    simd = ((hw >> 16) & 15) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 15) * 4 + ((hw >> 4) & 3)   # xcc, se, sh, cu, simd
    ids = np.unique(simd[work])
    walked = np.minimum(n, entered * 64) * work
    per_simd_entries = np.array([walked[work & (simd == i)].sum() for i in ids])
    per_simd_end = np.array([end[work & (simd == i)].max() for i in ids])
    per_simd_waves = np.array([(work & (simd == i)).sum() for i in ids])
    print('   SIMDs used %d; waves per SIMD mean %.1f max %d; walked entries per SIMD mean %.0f p90 %.0f max %.0f; last end per SIMD mean %.1f p10 %.1f p90 %.1f max %.1f us' % (
        len(ids), per_simd_waves.mean(), per_simd_waves.max(), per_simd_entries.mean(), np.percentile(per_simd_entries, 90), per_simd_entries.max(),
        per_simd_end.mean(), np.percentile(per_simd_end, 10), np.percentile(per_simd_end, 90), per_simd_end.max()))
    cc = np.corrcoef(per_simd_entries, per_simd_end)[0, 1]
    print('   correlation (walked entries on a SIMD, its last end) %.2f; ns per walked entry per SIMD: mean %.0f' % (cc, (per_simd_end * 1e3 / np.maximum(per_simd_entries, 1)).mean()))
    print('   first 24 launch indices -> (xcc, se, sh, cu, simd): ' + ' '.join('%d:%d.%d.%d.%d.%d' % (i, (hw[i] >> 16) & 15, (hw[i] >> 13) & 7, (hw[i] >> 12) & 1, (hw[i] >> 8) & 15, (hw[i] >> 4) & 3) for i in range(24)))
    print('   indices 1024..1031, 2048..2051: ' + ' '.join('%d:%d.%d.%d.%d.%d' % (i, (hw[i] >> 16) & 15, (hw[i] >> 13) & 7, (hw[i] >> 12) & 1, (hw[i] >> 8) & 15, (hw[i] >> 4) & 3) for i in list(range(1024, 1032)) + list(range(2048, 2052))))
    busy = dur[work].sum()
    print('   sum of wave durations %.0f us = %.1f waves busy on average over the span' % (busy, busy / end[work].max()))
