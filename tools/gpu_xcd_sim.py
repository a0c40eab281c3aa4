"""What the XCD-aware launch order of the forward blend buys and costs: for C3 ring views, the walked
entries per XCD (balance: the launch ends with its slowest XCD) and the splat records each XCD's L2 has to fetch (64 B per
distinct (record, XCD) pair among the walked entries) under (a) the launch order as built (block b -> XCD b mod 8), (b) one
length-sorted sequence (rounds 2-5), (c) other ways of dealing sub-tiles to XCDs.  Reads the workspaces of real renders.
python tools/gpu_xcd_sim.py [views...]"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import exavatar_release_amd as exa
from exavatar_release_amd import scenes, stats, rasterizer as rz
from exavatar_release_amd.camera import make_raster_matrices
dev = torch.device('cuda:0'); H = W = 1024; P = 150000
a = {k: v.to(dev).requires_grad_(True) for k, v in scenes.dist_b_avatar(P, seed=0).items()}
G = torch.randn(3, H, W, device=dev)
exa.config.mode = 'exact'; exa.config.keep_debug = True
views = [int(v) for v in sys.argv[1:]] or [0, 50, 100, 150]
for view in views:
    tanx, tany, vm, pm, cp = make_raster_matrices(scenes.ring_camera(H, W, view, 200), (H, W))
    st = exa.GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, vm.to(dev), pm.to(dev), 0, cp.to(dev), False, False)
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    color = rz.rasterize_gaussians(a['mean_3d'], m2, None, a['rgb'], a['opacity'], a['scale'], a['rotation'], None, st)[0]
    torch.cuda.synchronize()
    tile, binws, cap = rz._debug_last['tile'], rz._debug_last['bin'], int(rz._debug_last['capacity'])
    lay = stats.tile_offsets(P, W, H)
    nsub = lay['cells'] * 64
    rg = tile[lay['ranges'][0]:lay['ranges'][0] + nsub * 8].view(torch.int32).view(-1, 2).long()
    ex = tile[lay['fwd_exit'][0]:lay['fwd_exit'][0] + nsub * 8].view(torch.int32).view(-1, 2).long()
    slots = tile[lay['slots'][0]:lay['slots'][0] + nsub * 16].view(torch.int32).view(-1, 4).long()
    length = rg[:, 1] - rg[:, 0]
    walked = torch.where(length > 0, torch.minimum(ex[:, 0], ex[:, 1] * 64), torch.zeros_like(length))
    b = stats.bin_offsets(cap)['sorted'][0]
    ids = binws[b:b + cap * 4].view(torch.int32).long()
    # sub-tile geometry: st = cell * 64 + (sy * 8 + sx) inside the cell, cells row-major
    cx_n = (W + 63) // 64
    sub = torch.arange(nsub, device=dev)
    cell = sub // 64
    gsx = (cell % cx_n) * 8 + (sub % 64) % 8
    gsy = (cell // cx_n) * 8 + (sub % 64) // 8
    pos_of = torch.empty(nsub, dtype=torch.long, device=dev)
    pos_of[slots[:, 2]] = torch.arange(nsub, device=dev)
    cls = torch.where(length > 0, torch.clamp((length + 15) // 16, max=63), torch.zeros_like(length))
    one = torch.empty(nsub, dtype=torch.long, device=dev)
    one[torch.argsort(-cls, stable=True)] = torch.arange(nsub, device=dev)
    schemes = {'as built (position b -> XCD b mod 8)': pos_of % 8,
               'one length-sorted sequence (rounds 2-5)': one % 8,
               'cells hashed (cx + 3 cy) mod 8': ((cell % cx_n) + 3 * (cell // cx_n)) % 8,
               '4x4 blocks hashed (bx + 3 by) mod 8': ((gsx // 4) + 3 * (gsy // 4)) % 8,
               '2x2 blocks hashed': ((gsx // 2) + 3 * (gsy // 2)) % 8}
    # walked entries as (id, sub-tile) pairs
    nz = torch.nonzero(walked > 0).flatten()
    reps = walked[nz]
    owner = torch.repeat_interleave(nz, reps)
    start = torch.repeat_interleave(rg[nz, 0], reps)
    off = torch.arange(int(reps.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(reps, 0) - reps, reps)
    eid = ids[start + off]
    print('view %d: %d lists, %d walked entries (%.1f MB of record gathers without any L2 reuse), %d distinct records' % (
        view, int((length > 0).sum()), int(reps.sum()), int(reps.sum()) * 64 / 1e6, int(torch.unique(eid).numel())))
    for name, xcd in schemes.items():
        per = torch.zeros(8, device=dev).index_add_(0, xcd, walked.float())
        uniq = torch.unique(eid * 8 + xcd[owner]).numel()
        print('   %-38s walked per XCD max / mean = %.3f   record fetches %.1f MB' % (name, float(per.max() / per.mean()), uniq * 64 / 1e6))
