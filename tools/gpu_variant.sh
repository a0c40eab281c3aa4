#!/bin/bash
# A/B a variant build of the library (same ABI, EXA_RASTER_LIB): parity smoke + bench line + kernel times.
# Usage on the GPU box: bash tools/gpu_variant.sh gpurun_variants/libexa_X.so
V=$GRAFT_REPO_ROOT/$1
cd $GRAFT_REPO_ROOT
echo "== variant $1"
EXA_RASTER_LIB=$V timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
EXA_RASTER_LIB=$V timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})
for k in ('extra_batched_views', 'extra_batched_views_x2', 'extra_views_in_flight'):
    print(k, round(d[k]['value'], 1))"
