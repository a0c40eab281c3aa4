#!/bin/bash
# Round-4 check U: the view switch of the replayed step as the first node of the graph (exa_raster_select_row) against the
# eager elementwise kernel in front of every replay.
R=$GRAFT_REPO_ROOT; cd $R
ab() {
  env $1 timeout 200 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'], 1), round(d['ms_per_step'], 4), d['config']['view_switch'][:110])"
}
for i in 1 2 3; do ab EXA_BENCH_CAM_COPY=graph; ab EXA_BENCH_CAM_COPY=kernel; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bench" 2>&1 | tail -2
