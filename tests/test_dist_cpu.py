"""world_size-2 gloo tests of the N > 1 path: view sharding, flat gradient all-reduce, densify stats."""
import os

import pytest
import torch
import torch.distributed as dist
import multiprocessing as mp

from exavatar_release_amd.dist import FlatGradAllReducer, reduce_densify_stats, shard_views


def test_shard_views_partition_and_reshuffle():
    for world in (1, 2, 3, 8):
        for epoch in (0, 1):
            shards = [shard_views(200, r, world, epoch=epoch, pad=False) for r in range(world)]
            allv = sorted(v for s in shards for v in s)
            assert allv == list(range(200))
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
            # default: padded to equal length (every rank issues the same number of all-reduces per epoch)
            padded = [shard_views(200, r, world, epoch=epoch) for r in range(world)]
            assert len({len(s) for s in padded}) == 1 and len(padded[0]) == -(-200 // world)
            assert set(v for s in padded for v in s) == set(range(200))
    assert shard_views(200, 0, 8, epoch=0) != shard_views(200, 0, 8, epoch=1)
    assert shard_views(10, 1, 2, shuffle=False) == [1, 3, 5, 7, 9]
    assert shard_views(10, 2, 3, shuffle=False) == [2, 5, 8, 1] and shard_views(10, 2, 3, shuffle=False, pad=False) == [2, 5, 8]
    # an explicit order (bench.py's stratified ring order) is dealt like the seeded permutation
    order = [(i * 123) % 200 for i in range(200)]
    dealt = [shard_views(200, r, 8, order=order) for r in range(8)]
    assert sorted(v for s_ in dealt for v in s_) == list(range(200)) and dealt[3][:3] == [order[3], order[11], order[19]]
    with pytest.raises(ValueError):
        shard_views(4, 0, 2, order=[0, 1, 1, 3])
    # fewer views than ranks: the permutation is repeated, no rank is left with an empty shard (it would hang the others)
    assert [shard_views(1, r, 4, shuffle=False) for r in range(4)] == [[0]] * 4
    assert [len(shard_views(3, r, 8)) for r in range(8)] == [1] * 8


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        P = 1000
        g = torch.Generator().manual_seed(100 + rank)
        shapes = [(P, 3), (P, 3), (P, 4), (P, 1), (P, 3)]
        grads = [torch.randn(*s, generator=g) for s in shapes]
        red = FlatGradAllReducer(grads, average=True, n_buffers=2)
        assert red.nbytes == P * 14 * 4
        red.start(grads)
        # double buffering: a second step may pack while the first collective is in flight; both results survive
        red.start([2.0 * g_ for g_ in grads])
        red.finish()
        assert torch.allclose(red.buffer_views(1)[0], 2.0 * red.buffer_views(0)[0], atol=1e-6)
        out = [v.clone() for v in red.buffer_views(0)]
        # second step: rank 1 did not touch the opacity parameter (None grad = zeros)
        grads2 = [torch.full(s, float(rank + 1)) for s in shapes]
        if rank == 1:
            grads2[3] = None
        red.start(grads2)
        out2 = [v.clone() for v in red.finish()]
        acc = torch.full((P,), float(rank + 1))
        cnt = torch.full((P,), 1.0 + rank)
        rmax = torch.arange(P, dtype=torch.float32) * (1 if rank == 0 else -1)
        reduce_densify_stats(acc, cnt, rmax)
        q.put((rank, [o.numpy() for o in out], [o.numpy() for o in out2], acc.numpy(), cnt.numpy(), rmax.numpy()))
    finally:
        dist.destroy_process_group()


def test_flat_grad_allreduce_world2_gloo():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, out, out2, acc, cnt, rmax = q.get(timeout=120)
        res[rank] = ([torch.from_numpy(o) for o in out], [torch.from_numpy(o) for o in out2],
                     torch.from_numpy(acc), torch.from_numpy(cnt), torch.from_numpy(rmax))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # expected mean of the two ranks' seeded gradients
    shapes = [(1000, 3), (1000, 3), (1000, 4), (1000, 1), (1000, 3)]
    exp = []
    for i, s in enumerate(shapes):
        gs = []
        for r in range(world):
            g = torch.Generator().manual_seed(100 + r)
            gg = [torch.randn(*ss, generator=g) for ss in shapes]
            gs.append(gg[i])
        exp.append((gs[0] + gs[1]) / 2)
    for r in range(world):
        out, out2, acc, cnt, rmax = res[r]
        for o, e in zip(out, exp):
            assert torch.allclose(o, e, atol=1e-6)
        assert torch.allclose(out2[0], torch.full((1000, 3), 1.5))
        assert torch.allclose(out2[3], torch.full((1000, 1), 0.5))       # (1 + 0) / 2
        assert torch.all(acc == 3.0) and torch.all(cnt == 3.0)
        assert torch.equal(rmax, torch.arange(1000, dtype=torch.float32))
    assert torch.equal(res[0][0][0], res[1][0][0])


def _worker8(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        P = 257
        shapes = [(P, 3), (P, 3), (P, 4), (P, 1), (P, 3)]
        like = [torch.zeros(s) for s in shapes]
        red = FlatGradAllReducer(like, average=False, n_buffers=2)
        sums = []
        # five double-buffered steps; in step s, rank r leaves tensor (r + s) % 5 untouched (None) and ranks with
        # r % 3 == s % 3 contribute nothing at all (a rank whose view saw no Gaussian): uneven None patterns must neither
        # hang the collective nor leave stale values of an earlier step in the buffer
        for step in range(5):
            grads = [torch.full(s, float((rank + 1) * (i + 1) + 10 * step)) for i, s in enumerate(shapes)]
            grads[(rank + step) % 5] = None
            if rank % 3 == step % 3:
                grads = [None] * 5
            red.start(grads)
            if step >= 1:
                b = (step - 1) % 2
                red.wait(b)
                sums.append([v.clone() for v in red.buffer_views(b)])
        red.finish()
        sums.append([v.clone() for v in red.buffer_views(4 % 2)])
        q.put((rank, [[float(t.flatten()[0]) for t in st] + [float((t - t.flatten()[0]).abs().max()) for t in st] for st in sums]))
    finally:
        dist.destroy_process_group()


def test_flat_grad_allreduce_world8_uneven_none_grads():
    """Eight ranks (gloo, CPU): the world size of the driver's scaling run.  Every step has a different pattern of missing
    gradients per rank; all ranks must see the same sums, equal to the closed form, for every one of the five
    double-buffered steps."""
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, vals = q.get(timeout=300)
        res[rank] = vals
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step in range(5):
        exp = []
        for i in range(5):
            tot = 0.0
            for r in range(world):
                if r % 3 == step % 3 or (r + step) % 5 == i:
                    continue
                tot += (r + 1) * (i + 1) + 10 * step
            exp.append(tot)
        for r in range(world):
            got = res[r][step]
            assert got[:5] == exp, (step, r, got[:5], exp)
            assert max(got[5:]) == 0.0          # constant tensors: no stale element anywhere
