#!/bin/bash
# A/B of library variants (exavatar_release_amd/_variants/<name>.so, tools/build_variant.sh) against the in-tree build:
# bash tools/gpu_ab_variants.sh <name> [<name> ...]; the bench line of each, twice, alternating.
R=$GRAFT_REPO_ROOT; cd $R
ab() {
  echo "== $*"
  env "$@" timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
}
for i in 1 2; do
  ab EXA_X=0
  for v in "$@"; do ab EXA_RASTER_LIB=exavatar_release_amd/_variants/$v.so; done
done
