#!/bin/bash
# Developer driver for one gpurun call: smoke, the GPU test suite, the bench line, per-kernel times.
# Usage (on the GPU box): bash tools/gpu_round.sh <tag> [notest]
TAG=${1:-r02a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.log
tail -2 $O/smoke.log
if [ "$2" != "notest" ]; then
timeout 420 python -m pytest tests -m gpu -q --timeout 150 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
grep -E "^E  |FAILED|passed|failed" $O/pytest.log | head -40
fi
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.log
cat $O/bench.json; tail -5 $O/bench.err
timeout 300 python tools/gpu_kernel_times.py 0 50 > $O/ktimes.log 2>&1; cat $O/ktimes.log
timeout 300 python tools/gpu_batch_times.py > $O/btimes.log 2>&1; cat $O/btimes.log
