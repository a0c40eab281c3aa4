#!/bin/bash
# C3 bench line, C5 forward and the five-render iteration for library variants (in-tree build = "intree").
R=$GRAFT_REPO_ROOT; cd $R
for v in "$@"; do
  L=""; [ $v != intree ] && L="EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/$v.so"
  echo "== $v"
  env $L timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  c3', round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
  env $L timeout 200 python bench.py --config c5 --mode forward --steps 200 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs --no-kernel-timing 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  c5', round(d['value'], 1), round(d['ms_per_step'], 4))"
done
bash tools/gpu_iter_ab.sh "$@" 2>&1 | grep -A1 "^==" | grep -v "^--" | sed 's/, preprocess_bwd.*//' | cut -c1-260
