#!/bin/bash
# Round-4 check I: render_bwd prologue variants, identical otherwise: vA = flat gradient pointer + round 3's payload order,
# vB = global / scalar pointer, vC = + all payload loads in one round trip.
R=$GRAFT_REPO_ROOT; cd $R
ab() {
  env EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/$1.so timeout 200 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'], 1), round(d['ms_per_step'], 4), 'bwd', round(d['roofline']['kernel_avg_us']['render_bwd'], 2))"
}
for i in 1 2 3; do ab vA; ab vB; ab vC; done
