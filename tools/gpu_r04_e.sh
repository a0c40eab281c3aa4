#!/bin/bash
# Round-4 check E: per-kernel profile (rocprofv3 --stats) of the graphed five-render iteration; iteration bench.
TAG=${1:-r04e}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_gpu_graphed_iteration.py tests/test_gpu_edge_cases.py tests/test_gpu_parity.py -q --timeout 250 -k "graphed or composite or iteration or compose" 2>&1 | tail -12 | cut -c1-300
timeout 300 python - <<PY 2>&1 | grep -v amdgpu.ids | tee $O/iter_bench.log
import torch, bench, json
r = bench.iteration_throughput(torch.device('cuda:0'), iters=30)
for k in ('sets', 'graphed', 'graphed_photometric'):
    print(k, json.dumps(r[k])[:160])
PY
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iter_graphed -- python $R/tools/gpu_iteration_profile.py graphed 60 > $O/iter_graphed.log 2>&1
tail -1 $O/iter_graphed.log
find $O/iter_graphed -name "*kernel_stats.csv" | head -1 | xargs head -40 | cut -c1-170
