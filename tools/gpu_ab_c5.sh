#!/bin/bash
# A/B of library variants on config C5 (forward only, 2048 x 2048, SH 3): bash tools/gpu_ab_c5.sh <variant> ...
R=$GRAFT_REPO_ROOT; cd $R
ab() {
  echo "== $*"
  env "$@" timeout 200 python bench.py --config c5 --mode forward --steps 200 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs --no-kernel-timing 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4))"
}
for i in 1 2; do
  ab EXA_X=0
  for v in "$@"; do ab EXA_RASTER_LIB=exavatar_release_amd/_variants/$v.so; done
done
