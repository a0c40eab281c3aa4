"""The launch orders the sort launch leaves for the two blends (csrc/render_fwd.hip order_slots), read back from the workspaces
of real renders: eight interleaved length-sorted streams, one per XCD region (tests/test_bwd_order_model.py is the arithmetic)."""
import numpy as np
import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import rasterizer as rz, scenes, stats
from exavatar_release_amd.camera import make_raster_matrices

pytestmark = pytest.mark.gpu
BATCH, DEPTH, MAGIC = 64, 16, 0xB07DE7ED


def _region(local):
    return ((local >> 1) & 3) | ((local >> 3) & 4)


def _render(P, H, W, view, scene):
    dev = torch.device('cuda:0')
    a = {k: v.to(dev).requires_grad_(True) for k, v in scene.items()}
    tanx, tany, vm, pm, cp = make_raster_matrices(scenes.ring_camera(H, W, view, 200), (H, W))
    st = exa.GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3, device=dev), 1.0, vm.to(dev), pm.to(dev), 0, cp.to(dev),
                                           False, False)
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    out = rz.rasterize_gaussians(a['mean_3d'], m2, None, a['rgb'], a['opacity'], a['scale'], a['rotation'], None, st)
    torch.cuda.synchronize()
    dbg = rz._debug_last
    return out, dbg['tile'].cpu().numpy(), dbg['bin'].cpu().numpy(), int(dbg['capacity'])


@pytest.mark.parametrize('case', ['avatar_1024', 'avatar_small', 'two_splats'])
def test_launch_orders_are_interleaved_region_streams(case):
    old = (exa.config.mode, exa.config.keep_debug)
    exa.config.mode, exa.config.keep_debug = 'exact', True
    try:
        if case == 'avatar_1024':
            P, H, W = 150000, 1024, 1024
            scene = scenes.dist_b_avatar(P, seed=0)
        elif case == 'avatar_small':
            P, H, W = 20000, 200, 328                       # border cells with padding sub-tiles
            scene = scenes.dist_b_avatar(P, seed=1)
        else:
            P, H, W = 2, 256, 256                           # seven of the eight streams (nearly) empty
            scene = scenes.dist_b_avatar(64, seed=2)
            scene = {k: v[:2].clone() for k, v in scene.items()}
        out, tile, binws, cap = _render(P, H, W, 0, scene)
    finally:
        exa.config.mode, exa.config.keep_debug = old
    lay = stats.tile_offsets(P, W, H)
    nsub = lay['cells'] * 64
    rg = tile[lay['ranges'][0]:lay['ranges'][0] + nsub * 8].view(np.uint32).reshape(-1, 2).astype(np.int64)
    slots = tile[lay['slots'][0]:lay['slots'][0] + nsub * 16].view(np.uint32).reshape(-1, 4).astype(np.int64)
    meta = tile[128:144].view(np.uint32).astype(np.int64)
    n = rg[:, 1] - rg[:, 0]
    cls = np.where(n > 0, np.minimum(63, (n + 15) // 16), 0)
    reg = _region(np.arange(nsub) & 63)
    # forward: a permutation, position mod 8 = region, the record carries the list's range, descending class per stream
    assert sorted(slots[:, 2].tolist()) == list(range(nsub))
    pos = np.arange(nsub)
    assert np.array_equal(reg[slots[:, 2]], pos % 8)
    assert np.array_equal(slots[:, 0], rg[slots[:, 2], 0]) and np.array_equal(slots[:, 1], rg[slots[:, 2], 1])
    for x in range(8):
        seq = cls[slots[x::8, 2]]
        ne = seq[seq > 0]
        assert np.all(np.diff(ne) <= 0) and np.all(seq[len(ne):] == 0)
    # backward: every batch of every list exactly once; all eight streams alive => position mod 8 = region
    assert meta[2] == MAGIC
    total = int(meta[0] + meta[1])
    b0 = stats.bin_offsets(cap)['bucket'][0]
    order = binws[b0:b0 + total * 4].view(np.uint32).astype(np.int64)
    expect = np.concatenate([rg[i, 0] // BATCH + np.arange((n[i] + BATCH - 1) // BATCH) for i in np.flatnonzero(n > 0)] or
                            [np.zeros(0, dtype=np.int64)])
    assert sorted(order.tolist()) == sorted(expect.tolist())
    slot_region = {}
    for i in np.flatnonzero(n > 0):
        for b in range((n[i] + BATCH - 1) // BATCH):
            slot_region[rg[i, 0] // BATCH + b] = (reg[i], b, cls[i])
    M = np.zeros(8, dtype=np.int64)
    for s in order[:meta[0]]:
        M[slot_region[s][0]] += 1
    aligned = 8 * int(M.min())
    assert all(slot_region[s][0] == q % 8 for q, s in enumerate(order[:aligned].tolist()))
    for x in range(8):              # every stream batch-major, heavy lists first
        seq = [slot_region[s] for s in order[:meta[0]] if slot_region[s][0] == x]
        keys = [(b, -c) for _, b, c in seq]
        assert keys == sorted(keys)
        assert all(b < DEPTH for _, b, _ in seq)
    if case == 'avatar_1024':       # the streams of a real view are balanced: nearly everything stays aligned
        assert aligned >= 0.9 * meta[0], (M, meta)
