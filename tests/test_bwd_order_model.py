"""The launch orders of the two blends (csrc/render_fwd.hip order_slots, csrc/render_bwd.hip).  Workgroup b of a launch runs
on XCD b mod 8, so both orders are EIGHT interleaved streams: sub-tile st belongs to region xcd_region(st & 63) (blocks of 2 x 4
sub-tiles, one per region and cell); the k-th sub-tile of region x by descending list-length class
(class = min(63, ceil(n / 16)), empty lists last) is launched at position 8 k + x; batch b of that list is entry
kb = B_b + k of the region's backward stream, B_b = sum over b' < b of the number of the region's lists with more than b' batches
(more than b batches <=> class > 4 b), and sits at position sum over x' of min(M_x', kb) + #{x' < x: M_x' > kb} (M = the streams'
lengths): 8 kb + x while all eight streams are alive, compacted behind the end of the shortest; batches from the 17th on are
appended behind.  A numpy model of exactly that arithmetic: it must be a bijection onto [0, sum M) for any list lengths, batch-
major and longest list first inside every region.  (The kernels themselves are covered by tests/test_gpu_launch_order.py and
by every GPU parity test: a wrong order would skip or repeat batches and break every gradient.)"""
import numpy as np
import pytest

BATCH, DEPTH, CLASSES, REGIONS = 64, 16, 64, 8


def xcd_region(local):
    return ((local >> 1) & 3) | ((local >> 3) & 4)


def order_model(n, rng):
    """n: list length of every sub-tile (cell-major, 64 per cell).  Returns the forward positions, the main part of the
    backward order {position: (region, stream entry, sub-tile, batch)}, the appended tail and the per-region stream lengths."""
    st = np.arange(len(n))
    reg = xcd_region(st & 63)
    cls = np.where(n > 0, np.minimum(CLASSES - 1, (n + 15) // 16), 0)
    fwd_pos = np.empty(len(n), dtype=np.int64)
    stream, tail, M = [], [], np.zeros(REGIONS, dtype=np.int64)
    for x in range(REGIONS):
        mine = reg == x
        hist = np.bincount(cls[mine], minlength=CLASSES)
        # s_off[c] = the region's lists of a class above c; class 0 (empty) comes last: everything non-empty is before it
        s_off = np.array([hist[c + 1:].sum() for c in range(CLASSES)])
        s_off[0] = hist[1:].sum()
        k = np.empty(len(n), dtype=np.int64)
        for c in range(CLASSES):
            idx = np.flatnonzero(mine & (cls == c))
            rng.shuffle(idx)                      # the kernel hands ranks out with atomics: arbitrary inside a class
            k[idx] = s_off[c] + np.arange(len(idx))
        fwd_pos[mine] = 8 * k[mine] + x
        bbase = np.concatenate(([0], np.cumsum([s_off[4 * b] for b in range(DEPTH)])))
        M[x] = bbase[DEPTH]
        for i in np.flatnonzero(mine & (n > 0)):
            for b in range((n[i] + BATCH - 1) // BATCH):
                if b < DEPTH:
                    stream.append((x, bbase[b] + k[i], i, b))
                else:
                    tail.append((i, b))
    main = {}
    for x, kb, i, b in stream:
        q = int(np.minimum(M, kb).sum() + (M[:x] > kb).sum())
        assert q not in main, 'two batches at one position'
        main[q] = (x, kb, i, b)
    return fwd_pos, main, tail, M, reg, cls


def test_regions_deal_every_cell_evenly():
    reg = xcd_region(np.arange(64))
    assert np.bincount(reg, minlength=8).tolist() == [8] * 8
    grid = reg.reshape(8, 8)                       # [sy][sx]
    for sy in range(0, 8, 4):
        for sx in range(0, 8, 2):
            assert len(set(grid[sy:sy + 4, sx:sx + 2].ravel().tolist())) == 1, 'a block of 2 x 4 sub-tiles is one region'


@pytest.mark.parametrize('seed', range(8))
def test_interleaved_streams_are_a_bijection(seed):
    rng = np.random.default_rng(seed)
    kind = seed % 4
    if kind == 0:       # avatar-like: many empty sub-tiles, lists of 1 .. 700
        n = np.where(rng.random(4096) < 0.25, rng.integers(1, 700, 4096), 0)
    elif kind == 1:     # a few very long lists (> 1024 entries: the appended tail), lengths at the class / batch boundaries
        n = rng.choice([0, 1, 15, 16, 17, 63, 64, 65, 127, 128, 129, 1007, 1008, 1009, 1024, 1025, 3000], 2048)
    elif kind == 2:     # content everywhere
        n = rng.integers(1, 1500, 1024)
    else:               # next to nothing, in one corner: seven streams are empty
        n = np.zeros(256, dtype=np.int64)
        n[:2] = (700, 3)
    fwd_pos, main, tail, M, reg, cls = order_model(n, rng)
    # forward: a permutation whose position mod 8 is the region, descending class along every stream
    assert sorted(fwd_pos.tolist()) == list(range(len(n)))
    assert np.array_equal(fwd_pos % 8, reg)
    for x in range(REGIONS):
        mine = np.flatnonzero(reg == x)
        seq = cls[mine][np.argsort(fwd_pos[mine])]
        ne = seq[seq > 0]
        assert np.all(np.diff(ne) <= 0) and np.all(seq[len(ne):] == 0)
    # backward: a bijection onto [0, sum M); while all streams are alive position q runs on XCD q mod 8 = the region
    assert sorted(main) == list(range(int(M.sum())))
    for q, (x, kb, i, b) in main.items():
        assert reg[i] == x
        if kb < M.min():
            assert q == 8 * kb + x
    batches = sum((v + BATCH - 1) // BATCH for v in n)
    assert len(main) + len(tail) == batches
    assert all(b >= DEPTH and n[i] > DEPTH * BATCH for i, b in tail)
    for x in range(REGIONS):        # every stream in launch order: batch-major, inside one batch index descending class
        seq = [main[q] for q in sorted(main) if main[q][0] == x]
        assert [kb for _, kb, _, _ in seq] == list(range(M[x]))
        prev = (-1, CLASSES)
        for _, _, i, b in seq:
            assert (b, -cls[i]) >= (prev[0], -prev[1])
            prev = (b, cls[i])
    # the launch has one wave per batch slot in use = batches + one end slot per list: the order always fits
    assert int(M.sum()) + len(tail) == batches
