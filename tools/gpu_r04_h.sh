#!/bin/bash
# Round-4 check H: A/B of the working build against HEAD (r04o.so).
TAG=${1:-r04h}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.log; tail -4 $O/pytest.log | cut -c1-300
ab() {
  echo "== $1 $2"
  env $1 timeout 200 python bench.py $2 --steps 300 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
}
ab EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/r04o.so
ab EXA_X=0
ab EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/r04o.so
ab EXA_X=0
ab EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/r04o.so "--config c5"
ab EXA_X=0 "--config c5"
ab EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/r04o.so "--config c2"
ab EXA_X=0 "--config c2"
