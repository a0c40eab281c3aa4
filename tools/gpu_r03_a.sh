#!/bin/bash
# Round-3 check A (one gpurun call): full GPU suite, default bench line, host profile of the eager drop-in path,
# GraphedRenderer frame times, rocprofv3 kernel statistics of the bench command.
# Usage on the GPU box: bash tools/gpu_r03_a.sh <tag>
TAG=${1:-r03a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout 300 --durations=8 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.log; tail -25 $O/pytest.log | cut -c1-300
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.log
python - <<PY
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(round(d['value'], 1), d['ms_per_step'], {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})
print('frac', d['roofline']['frac'], 'step frac', d['roofline']['step']['frac_at_measured_step'], 'D', d['config']['mean_instances_D'], d['config']['mean_subtile_instances'])
print('cpu', d.get('cpu_baseline'))
print('iteration', d.get('extra_exavatar_iteration'))
for k in ('extra_batched_views', 'extra_batched_views_x2', 'extra_views_in_flight', 'roofline_batched', 'extra_c5_forward', 'extra_c2', 'rccl_world1_smoke'):
    print(k, d.get(k))
PY
tail -3 $O/bench.err
timeout 200 python tools/gpu_host_profile.py > $O/host_profile.log 2>&1; head -3 $O/host_profile.log; sed -n 4,40p $O/host_profile.log | cut -c1-200
timeout 300 python tools/gpu_graphed_times.py > $O/graphed.log 2>&1; cat $O/graphed.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 100 --warmup 10 \
  --no-cpu-baseline --no-concurrent --no-other-configs > $O/stats_bench.json 2> $O/stats.err
echo "rocprof rc=$?" | tee -a $O/summary.log
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -14 | cut -c1-160
