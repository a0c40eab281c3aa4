#!/bin/bash
# Round-4 check M: loss-in-graph tests again + per-kernel profile of the graphed iteration as it is now.
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_graphed_iteration.py -x -q 2>&1 | tail -30 | cut -c1-220
cd /tmp && export TMPDIR=/tmp
for how in graphed graphed_loss; do
  rm -rf /tmp/prof_$how
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$how -- python $R/tools/gpu_iteration_profile.py $how 60 > /tmp/prof_$how.log 2>&1
  tail -1 /tmp/prof_$how.log
  f=$(find /tmp/prof_$how -name '*kernel_stats.csv' | head -1)
  mkdir -p $R/gpurun_out/r04m; cp $f $R/gpurun_out/r04m/${how}_kernel_stats.csv
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows[:32]:
    n, avg = int(r['Calls']), float(r['AverageNs']) / 1e3
    tot += n * avg
    print('%-90s %5d %8.1f' % (r['Name'][:90], n, avg))
print('sum us per 70 iterations-ish:', tot, ' -> per iteration', tot / 70)
PY
done
