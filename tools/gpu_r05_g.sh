#!/bin/bash
# Round 5, run G: the final tree -- GPU suite in the driver's order, the same with poisoned workspaces, overflow stress, bench line.
mkdir -p gpurun_out/r05g
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r05g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05g/pytest.log
tail -3 gpurun_out/r05g/pytest.log
EXA_TEST_POISON=1 timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_bench_ranks.py > gpurun_out/r05g/pytest_poison.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05g/pytest_poison.log
tail -3 gpurun_out/r05g/pytest_poison.log
PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 600 python tools/gpu_overflow_stress.py 2 > gpurun_out/r05g/stress.log 2>&1; echo "stress rc=$?" >> gpurun_out/r05g/stress.log
timeout 600 python tools/gpu_overflow_stress.py 1 >> gpurun_out/r05g/stress.log 2>&1; echo "stress (caching allocator, iteration cases) rc=$?" >> gpurun_out/r05g/stress.log
grep "rc=\|STRESS" gpurun_out/r05g/stress.log
timeout 900 python bench.py > gpurun_out/r05g/bench.json 2> gpurun_out/r05g/bench.err; tail -c 300 gpurun_out/r05g/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05g/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['secondary']['frac'], d['roofline']['secondary']['counters_source'][:40])
print(json.dumps(d.get('extra_c3_lbs'))[:700])
PY
python -c "import __graft_entry__ as g; g.smoke()"
