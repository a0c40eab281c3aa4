#!/bin/bash
# A/B of library variants (EXA_RASTER_LIB): C3 (fwd + bwd) and C5 (forward) headline steps (default launch protocol of bench.py), plus the
# HIP-event kernel times.  Usage: bash tools/gpu_ab.sh [-n repeats] lib1.so lib2.so ...   (interleaved: lib1 lib2 lib1 lib2 ...)
cd $GRAFT_REPO_ROOT
N=2; if [ "$1" = -n ]; then N=$2; shift 2; fi
for i in $(seq $N); do
for lib in "$@"; do
  for c in c3 c5; do
    m=""; [ $c = c5 ] && m="--mode forward"
    EXA_RASTER_LIB=$GRAFT_REPO_ROOT/$lib timeout 250 python bench.py --config $c $m --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', '$c', round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
  done
done
done
