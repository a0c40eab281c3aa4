#!/bin/bash
# Round-3 check B (one gpurun call): full GPU suite, the backward parity tests again with the GC = 8 entry-major backward
# (EXA_BWD_GC=8), A/B of the bench step for backward variants, host profile, GraphedRenderer frame times.
TAG=${1:-r03b}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.log; tail -6 $O/pytest.log | cut -c1-300
EXA_BWD_GC=8 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py -m gpu -q --timeout 300 \
  -k "not bench and not footprint and not one_workgroup" > $O/pytest_gc8.log 2>&1
echo "pytest gc8 rc=$?" | tee -a $O/summary.log; tail -6 $O/pytest_gc8.log | cut -c1-300
ab() {
  echo "== $1"
  env $1 timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
}
ab EXA_X=0
ab EXA_BWD_GC=8
ab EXA_BWD_LDS_PAD=7000
ab EXA_BWD_LDS_PAD=3200
ab EXA_X=1
ab EXA_BWD_GC=8
timeout 300 python tools/gpu_host_profile.py > $O/host_profile.log 2>&1; head -3 $O/host_profile.log; grep -A 34 "Ordered by: internal" $O/host_profile.log | cut -c1-180; tail -3 $O/host_profile.log | cut -c1-600
timeout 300 python tools/gpu_graphed_times.py > $O/graphed.log 2>&1; cat $O/graphed.log | cut -c1-200
