"""Developer probe (library built with -DEXA_PROBE_SORT): cycles of the phases of every binsort_kernel workgroup."""
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
lay = tile_offsets(P, W, H); cells = lay['cells']
for k in (0, 50):
    tanx, tany, view, proj, cpos = make_raster_matrices(scenes.ring_camera(H, W, k, 200), (H, W))
    st = GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0, cpos.to(dev), False, False)
    m3, sc, rot, op, rgb = params
    with torch.no_grad():
        for _ in range(3):
            rasterize_gaussians(m3, torch.zeros(P, 3, device=dev), None, rgb, op, sc, rot, None, st)
    torch.cuda.synchronize()
    tile, binws, cap = _debug_last['tile'], _debug_last['bin'], int(_debug_last['capacity'])
    r = tile[lay['ranges'][0]: lay['ranges'][0] + lay['ranges'][1]].view(torch.int32).view(-1, 2).cpu().numpy().astype(np.int64)
    desc = tile[lay['cell_desc'][0]: lay['cell_desc'][0] + lay['cell_desc'][1]].view(torch.int32).view(-1, 4).cpu().numpy().astype(np.int64)
    n = (r[:, 1] - r[:, 0]).reshape(cells, 8, 8)
    nwg = cells * 8 + 16
    pr = binws[: cap * 8].view(torch.int32).cpu().numpy().astype(np.int64)[::-1][: 4 * nwg].reshape(nwg, 4)[:, ::-1] & 0xffffffff
    print('view', k, 'capacity', cap, 'order WGs total cycles:', pr[:16, 3].tolist())
    rows = []
    for rank in range(cells):
        cell = desc[rank, 0]; ents = desc[rank, 2] - desc[rank, 1]
        for row in range(8):
            tot = n[cell, row].sum()
            if tot:
                rows.append((pr[16 + rank * 8 + row, 3], pr[16 + rank * 8 + row].tolist(), int(ents), int(tot), n[cell, row].tolist()))
    rows.sort(key=lambda x: -x[0])
    print('  rows with keys: %d ; total-cycles mean %.0f p50 %.0f p90 %.0f max %.0f' % (len(rows), np.mean([x[0] for x in rows]), np.median([x[0] for x in rows]), np.percentile([x[0] for x in rows], 90), rows[0][0]))
    for x in rows[:3]:
        print('   top phases(ranges, scatter, wave sorts, end) %s entries %d keys %d lists %s' % (x[1], x[2], x[3], x[4]))
    for x in rows[len(rows) // 2: len(rows) // 2 + 4]:
        print('   (median) %s entries %d keys %d lists %s' % (x[1], x[2], x[3], x[4]))
