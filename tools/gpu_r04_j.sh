#!/bin/bash
# Round-4 check J: the view switch of bench.py as an elementwise kernel instead of the runtime's blit copy.
R=$GRAFT_REPO_ROOT; cd $R
ab() {
  env $1 timeout 200 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'], 1), round(d['ms_per_step'], 4))"
}
for i in 1 2 3; do ab EXA_BENCH_CAM_COPY=memcpy; ab EXA_BENCH_CAM_COPY=kernel; done
