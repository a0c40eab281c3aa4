#!/bin/bash
# Round 5, run B: single overflow protocol -- stress (fixed lib), GPU suite plain and with poisoned workspaces, short bench.
mkdir -p gpurun_out/r05b
PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 900 python tools/gpu_overflow_stress.py 2 > gpurun_out/r05b/stress.log 2>&1; echo "stress rc=$?" >> gpurun_out/r05b/stress.log
tail -3 gpurun_out/r05b/stress.log
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r05b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05b/pytest.log
tail -4 gpurun_out/r05b/pytest.log
EXA_TEST_POISON=1 timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_bench_ranks.py > gpurun_out/r05b/pytest_poison.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05b/pytest_poison.log
tail -4 gpurun_out/r05b/pytest_poison.log
timeout 600 python bench.py --no-other-configs --no-cpu-baseline > gpurun_out/r05b/bench.json 2> gpurun_out/r05b/bench.err; tail -c 600 gpurun_out/r05b/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05b/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_us'])
PY
