#!/bin/bash
# Round-3 check G: quick loop: edge-case tests, A/B against the library of HEAD.
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_edge_cases.py -m gpu -q --timeout 300 -x 2>&1 | tail -2 | cut -c1-300
ab() {
  echo "== $*"
  env "$@" timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
}
ab EXA_FUSED_BINSORT=0
ab EXA_FUSED_BINSORT=0 EXA_RASTER_LIB=exavatar_release_amd/_variants/head.so
ab EXA_FUSED_BINSORT=1
ab EXA_FUSED_BINSORT=0
ab EXA_FUSED_BINSORT=0 EXA_RASTER_LIB=exavatar_release_amd/_variants/head.so
