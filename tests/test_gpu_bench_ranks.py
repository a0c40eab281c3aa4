"""The N > 1 code path of bench.py with several ranks sharing the one GPU of the test box (gloo; RCCL refuses duplicate
devices).  Named so that it collects BEFORE the other GPU test files: SURVEY.md 8(e) is only ever exercised on hardware by
these two tests, and a failure later in the suite must not hide them (round 4: the suite aborted before reaching them)."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_bench_multi_rank_code_path_on_one_gpu():
    """`python bench.py --gpus 2 ...` AS TYPED (no torch.distributed.run in front: bench.py re-launches itself with one rank
    per GPU) -- the N > 1 path (view sharding, double-buffered flat gradients, async all-reduce around hipGraph replays,
    max-over-ranks timing) with two ranks sharing this GPU over gloo: RCCL itself refuses duplicate devices, and a 1-GPU
    box is all the tests get.  (The VALUES of the reduced gradient are checked in tests/test_gpu_edge_cases.py.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EXA_BENCH_BACKEND='gloo', MASTER_PORT='29541')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2',
           '--config', 'c2', '--no-kernel-timing', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    res = json.loads(line)
    assert res['n_gpus'] == 2 and res['steps'] == 6 and res['value'] > 0 and res['scaling'] == 'weak'
    assert res['rccl']['world_size'] == 2 and res['rccl']['backend'] == 'gloo'


def test_bench_two_ranks_with_views_in_flight():
    """`--views-in-flight 2` on the N > 1 path: groups of two views on two render slots, their gradients accumulated in slot
    order into the all-reducer's buffer, ONE all-reduce per group -- and the drop-in surface reported next to the headline."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EXA_BENCH_BACKEND='gloo', MASTER_PORT='29544', EXA_BENCH_SETTLE_STEPS='8')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2', '--config', 'c2',
           '--views-in-flight', '2', '--no-kernel-timing', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert res['n_gpus'] == 2 and res['steps'] == 5 and res['value'] > 0
    assert res['config']['views_in_flight_per_gpu'] == 2 and res['config']['views_per_step'] == 2
    assert all(rk['step_ms'] > 0 and rk['step_ms_without_allreduce'] > 0 for rk in res['rccl']['ranks'])
    assert res['extra_plugin_surface_eager']['value'] > 0


def test_bench_eight_ranks_on_one_gpu_as_the_driver_types_it():
    """`python bench.py --gpus 8 --steps K --warmup W` VERBATIM -- the command of the driver's 8-GPU scaling run -- with eight
    ranks sharing this GPU over gloo (RCCL refuses duplicate devices; EXA_BENCH_BACKEND is the only difference to the real
    run): the self-launch, the deal of 25 ring views per rank, two launch contexts with double-buffered flat gradients, the
    asynchronous all-reduce around the hipGraph replays, `finish()`, both barriers and the max-over-ranks timing all run
    to completion and rank 0 prints one line for a world of eight.  No scaling number is claimed from this."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EXA_BENCH_BACKEND='gloo', MASTER_PORT='29547', EXA_BENCH_SETTLE_STEPS='16')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '10', '--warmup', '3']
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    res = json.loads(line)
    assert res['n_gpus'] == 8 and res['steps'] == 10 and res['value'] > 0 and res['scaling'] == 'weak'
    assert res['rccl']['world_size'] == 8 and res['rccl']['backend'] == 'gloo'
    assert [r['rank'] for r in res['rccl']['ranks']] == list(range(8)) and all(r['world_size'] == 8 and r['views'] == 25 for r in res['rccl']['ranks'])
    assert res['config']['views_per_rank'] == 25 and res['config']['launch'] == 'abi'
    assert 'dp8' in res['config']['parallelism']
