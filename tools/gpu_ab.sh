#!/bin/bash
# A/B of library variants (EXA_RASTER_LIB): C3 and C5 bench lines.  Usage: bash tools/gpu_ab.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  echo "== $lib"
  for c in c3 c5; do
    m=""; [ $c = c5 ] && m="--mode forward"
    EXA_RASTER_LIB=$GRAFT_REPO_ROOT/$lib timeout 250 python bench.py --config $c $m --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$c', round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
  done
done
