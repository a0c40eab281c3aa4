#!/bin/bash
# Round-3 check C (one gpurun call): full GPU suite on the fused count+bin build, A/B against the previous binning and
# forward-blend experiment builds (exavatar_release_amd/_variants/*.so via EXA_RASTER_LIB), host profile, graphed times.
TAG=${1:-r03c}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.log; tail -8 $O/pytest.log | cut -c1-300
ab() {
  echo "== $1 $2"
  env $1 $2 timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
}
V=$R/exavatar_release_amd/_variants
ab EXA_X=0
ab EXA_RASTER_LIB=$V/prev_binning.so EXA_BWD_GC=8
ab EXA_RASTER_LIB=$V/fwd_noskip.so
ab EXA_RASTER_LIB=$V/fwd_exit8.so
ab EXA_RASTER_LIB=$V/fwd_both.so
ab EXA_X=1
ab EXA_RASTER_LIB=$V/prev_binning.so EXA_BWD_GC=8
echo "== c5 new / prev"
for lib in "" $V/prev_binning.so; do
  EXA_RASTER_LIB=$lib timeout 200 python bench.py --config c5 --steps 200 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
done
timeout 300 python tools/gpu_host_profile.py > $O/host_profile.log 2>&1; head -3 $O/host_profile.log | tail -2; tail -3 $O/host_profile.log | cut -c1-900
timeout 300 python tools/gpu_graphed_times.py > $O/graphed.log 2>&1; cat $O/graphed.log | cut -c1-200
