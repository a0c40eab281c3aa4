#!/bin/bash
# Quick A/B of the working tree's library against exavatar_release_amd/_variants/head.so (tools/build_variant.sh HEAD head):
# GPU tests first (TESTS = files / -k expression, default the edge cases), then the bench line, alternating.
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest ${TESTS:-tests/test_gpu_edge_cases.py} -m gpu -q --timeout 300 -x 2>&1 | tail -2 | cut -c1-300
ab() {
  echo "== $*"
  env "$@" timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
}
for i in 1 2 3; do
ab EXA_X=0
ab EXA_RASTER_LIB=exavatar_release_amd/_variants/head.so
done
