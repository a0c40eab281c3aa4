#!/bin/bash
# Headline + batched extras (K = 8 views per launch, two launches in flight, four views in flight) for library variants.
R=$GRAFT_REPO_ROOT; cd $R
for v in "$@"; do
  L=""; [ $v != intree ] && L="EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/$v.so"
  env $L timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-kernel-timing 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value'], 1), 'K8', round(d['extra_batched_views']['value'], 1), 'K8x2', round(d['extra_batched_views_x2']['value'], 1), 'inflight4', round(d['extra_views_in_flight']['value'], 1), 'iteration', {k: round(x['ms_per_iteration'], 3) for k, x in d['extra_exavatar_iteration'].items() if isinstance(x, dict)})"
done
