#!/bin/bash
# Final check of a round, part A (one gpurun call): the tests added this round, the default bench line, rocprofv3
# kernel statistics of the bench command.  Usage on the GPU box: bash tools/gpu_final_a.sh <tag>
TAG=${1:-r02z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 150 python -m pytest tests/test_gpu_parity.py -q --timeout 100 \
  -k "constant_prefix or render_iteration or clamped or render_many or batched_views" > $O/pytest_new.log 2>&1
echo "pytest_new rc=$?" | tee -a $O/summary.log; tail -4 $O/pytest_new.log | cut -c1-250
timeout 200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.log
python - <<PY
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(round(d['value'], 1), d['ms_per_step'], {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})
print('frac', d['roofline']['frac'], 'cpu', d.get('cpu_baseline'))
print('iteration', d.get('extra_exavatar_iteration'))
for k in ('extra_batched_views', 'extra_batched_views_x2', 'extra_views_in_flight', 'rccl_world1_smoke'):
    print(k, d.get(k))
PY
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 100 --warmup 10 \
  --no-cpu-baseline --no-concurrent > $O/stats_bench.json 2> $O/stats.err
echo "rocprof rc=$?" | tee -a $O/summary.log
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -14 | cut -c1-160
