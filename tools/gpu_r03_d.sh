#!/bin/bash
# Round-3 check D: full GPU suite, bench A/B (this build vs prev_binning variant), per-kernel profile of the five-render
# iteration (rocprofv3 --kernel-trace --stats), GraphedRenderer frame times.
TAG=${1:-r03d}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.log; tail -3 $O/pytest.log | cut -c1-300
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
ab EXA_X=1
timeout 300 python tools/gpu_graphed_times.py > $O/graphed.log 2>&1; cat $O/graphed.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for how in sets batched; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iter_$how -- python $R/tools/gpu_iteration_profile.py $how 40 > $O/iter_$how.log 2>&1
  tail -1 $O/iter_$how.log
  find $O/iter_$how -name "*kernel_stats.csv" | head -1 | xargs head -24 | cut -c1-140
done
