#!/bin/bash
# Round-4 check B: the new product tests (GraphedIteration, soak, overflow_check='forward'), then the whole GPU suite,
# A/B of the DPP asm reductions against the LDS-fix-only library.
TAG=${1:-r04b}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_graphed_iteration.py tests/test_gpu_soak.py -q --timeout 500 -x > $O/pytest_new.log 2>&1
echo "new rc=$?" | tee -a $O/summary.log; tail -40 $O/pytest_new.log | cut -c1-400
timeout 700 python -m pytest tests -m gpu -q --timeout 300 --deselect tests/test_gpu_graphed_iteration.py --deselect tests/test_gpu_soak.py > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.log; tail -15 $O/pytest.log | cut -c1-300
ab() {
  echo "== $1"
  env $1 timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
}
ab EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/r04lds.so
ab EXA_X=0
ab EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/r04lds.so
ab EXA_X=0
