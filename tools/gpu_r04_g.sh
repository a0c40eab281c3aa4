#!/bin/bash
# Round-4 check G: wave-cooperative gather in preprocess_bwd -- whole GPU suite, A/B against the previous library.
TAG=${1:-r04g}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.log; tail -6 $O/pytest.log | cut -c1-300
ab() {
  echo "== $1"
  env $1 timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
}
ab EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/r04h.so
ab EXA_X=0
ab EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/pbwd128.so
ab EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/r04h.so
ab EXA_X=0
ab EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/pbwd128.so
echo "== c5 fwd+bwd kernel times (scene splats: heavy path + stream)"
for lib in r04h.so ""; do
EXA_RASTER_LIB=${lib:+$R/exavatar_release_amd/_variants/$lib} timeout 200 python bench.py --config c5 --mode train --steps 40 --warmup 5 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
done
