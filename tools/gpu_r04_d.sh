#!/bin/bash
# Round-4 check D: GraphedIteration with indirect gradient pointers -- tests, host / device profile, the iteration bench.
TAG=${1:-r04d}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_graphed_iteration.py tests/test_gpu_soak.py -q --timeout 500 -x > $O/pytest_new.log 2>&1
echo "new rc=$?" | tee -a $O/summary.log; tail -25 $O/pytest_new.log | cut -c1-400
timeout 300 python tools/gpu_graphed_iter_profile.py 2>&1 | grep -v amdgpu.ids | tee $O/iter_profile.log
timeout 300 python - <<PY 2>&1 | grep -v amdgpu.ids | tee $O/iter_bench.log
import torch, bench, json
r = bench.iteration_throughput(torch.device('cuda:0'), iters=30)
for k, v in r.items():
    print(k, json.dumps(v)[:300])
PY
