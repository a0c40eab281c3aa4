#!/bin/bash
# Round-4 check F: waves per workgroup in render_bwd (EXA_BWD_WPB): composite backward (iteration) and the C3 headline.
TAG=${1:-r04f}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_graphed_iteration.py tests/test_gpu_edge_cases.py -q --timeout 250 -k "graphed or composite" 2>&1 | tail -3 | cut -c1-200
for w in 1 4; do
echo "== iteration EXA_BWD_WPB=$w (composites; plain backward too)"
EXA_BWD_WPB=$w timeout 300 python - <<PY 2>&1 | grep -v amdgpu.ids
import torch, bench, json
r = bench.iteration_throughput(torch.device('cuda:0'), iters=30)
for k in ('sets', 'graphed'):
    print(k, json.dumps(r[k])[:130])
PY
done
echo "== iteration default"
timeout 300 python - <<PY 2>&1 | grep -v amdgpu.ids
import torch, bench, json
r = bench.iteration_throughput(torch.device('cuda:0'), iters=30)
for k in ('sets', 'graphed'):
    print(k, json.dumps(r[k])[:130])
PY
ab() {
  echo "== $1"
  env $1 timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
}
ab EXA_BWD_WPB=1
ab EXA_BWD_WPB=2
ab EXA_BWD_WPB=4
ab EXA_BWD_WPB=1
ab EXA_BWD_WPB=2
