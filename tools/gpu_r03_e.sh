#!/bin/bash
# Round-3 check E: full GPU suite (composite renders), five-render iteration three ways + merged, per-kernel profile of it.
TAG=${1:-r03e}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.log; tail -30 $O/pytest.log | cut -c1-400
timeout 300 python - <<PY 2>&1 | tail -5 | cut -c1-1200
import torch, bench
print(bench.iteration_throughput(torch.device('cuda:0'), iters=30))
PY
ab() {
  echo "== $1 $2"
  env $1 $2 timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
}
ab EXA_X=0
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iter_sets -- python $R/tools/gpu_iteration_profile.py sets 40 > $O/iter_sets.log 2>&1
tail -1 $O/iter_sets.log
find $O/iter_sets -name "*kernel_stats.csv" | head -1 | xargs head -26 | cut -c1-140
