"""Developer probe (library built with -DEXA_PROBE_BWDLINE): start / end of every wave (= batch slot) of render_bwd on the
chip-wide 100 MHz clock against the number of blended entries of its batch.  C3, fwd + bwd, two ring views."""
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
params = [assets[k].to(dev).requires_grad_(True) for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')]
exa.config.mode = 'exact'; exa.config.keep_debug = True
lay = tile_offsets(P, W, H)
G = torch.randn(3, H, W, device=dev)
a256 = lambda v: (v + 255) & ~255
for k in [int(v) for v in (sys.argv[1:] or [0, 50])]:
    tanx, tany, view, proj, cpos = make_raster_matrices(scenes.ring_camera(H, W, k, 200), (H, W))
    st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0, cpos.to(dev), False, False)
    m3, sc, rot, op, rgb = params
    for _ in range(3):
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        out = rasterize_gaussians(m3, m2, None, rgb, op, sc, rot, None, st)
        (out[0] * G).sum().backward()
    torch.cuda.synchronize()
    tile, binws, cap = _debug_last['tile'], _debug_last['bin'], int(_debug_last['capacity'])
    nslots = cap // 64
    off_owner = a256(cap * 8) + a256(cap * 4) + a256(cap * 16)
    off_bmask = off_owner + a256((cap // 64 + 1) * 16)
    bm = binws[off_bmask: off_bmask + nslots * 8].view(torch.int64).cpu().numpy()
    owner = binws[off_owner: off_owner + nslots * 16].view(torch.int32).view(-1, 4).cpu().numpy()
    nb = np.array([bin(int(x) & 0xffffffffffffffff).count('1') for x in bm])
    tt = tile[lay['part_cnt'][0]: lay['part_cnt'][0] + lay['part_cnt'][1]].view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    m = min(nslots, len(tt) // 2)
    start, end = tt[0:2 * m:2] * 0.01, tt[1:2 * m:2] * 0.01
    # the probe records by WAVE (blockIdx); since round 3 wave w takes the w-th batch of the batch-major order the sort launch
    # left in the (dead) bucket array, and the waves past the order's end leave after one trip
    meta = tile[128:144].view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    off_bucket = a256(cap * 8) + a256(cap * 4)
    order = binws[off_bucket: off_bucket + nslots * 4].view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    if meta[2] == 0xB07DE7ED:
        n_ord = int(meta[0] + meta[1])
        slot_of = np.where(np.arange(m) < n_ord, np.minimum(order[:m], nslots - 1), -1)
        print('   batch-major order: %d + %d batches in the order (magic %#x)' % (meta[0], meta[1], meta[2]))
    else:
        slot_of = np.arange(m)
    nb = np.where(slot_of >= 0, nb[np.maximum(slot_of, 0)], 0)
    owner0 = np.where(slot_of >= 0, owner[np.maximum(slot_of, 0), 0], 0)
    work = (owner0 != 0) & (nb > 0)
    t0 = start[work].min(); start = start - t0; end = end - t0
    dur = end - start
    print('view %d: capacity %d = %d batch slots (%d probed), %d with blended entries (%d entries)' % (k, cap, nslots, m, work.sum(), nb[work].sum()))
    print('   starts of the working waves: p10 %.1f p50 %.1f p90 %.1f max %.1f us; ends: p50 %.1f p90 %.1f p99 %.1f max %.1f us' % (
        *[np.percentile(start[work], q) for q in (10, 50, 90, 100)], *[np.percentile(end[work], q) for q in (50, 90, 99, 100)]))
    for lo, hi in ((1, 8), (9, 16), (17, 32), (33, 48), (49, 64)):
        s_ = work & (nb >= lo) & (nb <= hi)
        if s_.any():
            print('   blended in [%2d,%2d]: %5d batches, duration mean %5.2f p90 %5.2f max %5.2f us' % (lo, hi, s_.sum(), dur[s_].mean(), np.percentile(dur[s_], 90), dur[s_].max()))
    idle = ~work
    print('   idle slots: %d, duration mean %.2f us; last idle start %.1f us' % (idle.sum(), dur[idle].mean(), start[idle].max()))
    busy = dur[work].sum()
    print('   sum of working-wave durations %.0f us = %.0f waves busy on average over %.1f us (chip: 1024 SIMDs)' % (busy, busy / end[work].max(), end[work].max()))
