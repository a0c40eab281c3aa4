#!/bin/bash
# Round-4 check T (experiment): boustrophedon launch order of the forward blend.
R=$GRAFT_REPO_ROOT; cd $R
ab() {
  env $1 timeout 200 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'], 1), round(d['ms_per_step'], 4), 'render_fwd', round(d['roofline']['kernel_avg_us']['render_fwd'], 2))"
}
for i in 1 2; do
  ab X=0
  for g in 1024 512 256; do ab EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/snake$g.so; done
done
