#!/bin/bash
# Round-4 check C: soak tests + the 8-rank bench command on one GPU, then the full default bench line.
TAG=${1:-r04c}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_soak.py "tests/test_gpu_parity.py::test_bench_eight_ranks_on_one_gpu_as_the_driver_types_it" -q --timeout 800 > $O/pytest_new.log 2>&1
echo "new rc=$?" | tee -a $O/summary.log; tail -40 $O/pytest_new.log | cut -c1-600
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.log
python - <<PY
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(round(d['value'], 1), d['ms_per_step'], 'no_settle', d.get('value_no_settle'), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})
print('frac', d['roofline']['frac'], 'step frac', d['roofline']['step']['frac_at_measured_step'], 'cpu', d.get('cpu_baseline'))
it = d.get('extra_exavatar_iteration', {})
for k in ('sequential', 'batched', 'sets', 'graphed'):
    print(k, it.get(k))
for k in ('extra_batched_views', 'extra_batched_views_x2', 'extra_views_in_flight', 'rccl_world1_smoke', 'extra_c5_forward', 'extra_c2', 'extra_cold_ring_order', 'roofline_batched'):
    print(k, d.get(k))
PY
tail -3 $O/bench.err
