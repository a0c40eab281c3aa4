"""The launch order of the backward blend (csrc/render_fwd.hip order_slots, csrc/render_bwd.hip): batch b of the list at
descending position p goes to position B_b + p, B_b = sum over b' < b of the number of lists with more than b' batches, taken
from the histogram of the list-length classes (class = min(63, ceil(n / 16)); more than b batches <=> class > 4 b).  A numpy
model of exactly that arithmetic: it must be a bijection onto [0, main) for any list lengths, batch-major, longest list
first, with the batches from the 17th on left to the appended tail.  (The kernels themselves are covered by the GPU parity
tests: a wrong order would skip or repeat batches and break every gradient.)"""
import numpy as np
import pytest

BATCH, DEPTH, CLASSES = 64, 16, 64


def order_model(n, rng):
    cls = np.where(n > 0, np.minimum(CLASSES - 1, (n + 15) // 16), 0)
    hist = np.bincount(cls, minlength=CLASSES)
    # s_off[c] = lists of a class above c (class 0 = empty lists, last in the order and not counted)
    s_off = np.array([hist[c + 1:].sum() for c in range(CLASSES)])
    s_off[0] = hist[1:].sum()
    # descending position: class by class, arbitrary order inside a class (the kernel hands ranks out with atomics)
    pos = np.empty(len(n), dtype=np.int64)
    for c in range(1, CLASSES):
        idx = np.flatnonzero(cls == c)
        rng.shuffle(idx)
        pos[idx] = s_off[c] + np.arange(len(idx))
    bbase = np.concatenate(([0], np.cumsum([s_off[4 * b] for b in range(DEPTH)])))
    main, tail = {}, []
    for i in np.flatnonzero(n > 0):
        nb = (n[i] + BATCH - 1) // BATCH
        for b in range(nb):
            if b < DEPTH:
                q = bbase[b] + pos[i]
                assert q not in main, 'two batches at one position'
                main[q] = (i, b)
            else:
                tail.append((i, b))
    return main, tail, bbase


@pytest.mark.parametrize('seed', range(6))
def test_batch_major_order_is_a_bijection(seed):
    rng = np.random.default_rng(seed)
    kind = seed % 3
    if kind == 0:       # avatar-like: many empty sub-tiles, lists of 1 .. 700
        n = np.where(rng.random(4096) < 0.25, rng.integers(1, 700, 4096), 0)
    elif kind == 1:     # a few very long lists (> 1024 entries: the appended tail), lengths at the class / batch boundaries
        n = rng.choice([0, 1, 15, 16, 17, 63, 64, 65, 127, 128, 129, 1007, 1008, 1009, 1024, 1025, 3000], 2048)
    else:               # content everywhere
        n = rng.integers(1, 1500, 1024)
    main, tail, bbase = order_model(n, rng)
    total_main = bbase[DEPTH]
    assert sorted(main) == list(range(total_main)), 'positions must cover [0, main) exactly once'
    batches = sum((v + BATCH - 1) // BATCH for v in n)
    assert total_main + len(tail) == batches
    assert all(b >= DEPTH and n[i] > DEPTH * BATCH for i, b in tail)
    # batch-major, and inside one batch index the lists in descending class order (heavy first)
    cls = np.minimum(CLASSES - 1, (n + 15) // 16)
    prev = (-1, CLASSES)
    for q in range(total_main):
        i, b = main[q]
        assert (b, -cls[i]) >= (prev[0], -prev[1])
        prev = (b, cls[i])
