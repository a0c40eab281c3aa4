#!/bin/bash
# Round profile: default bench line, rocprofv3 kernel stats of the bench command, HBM-traffic and SQ PMC passes
# (counters only with --kernel-trace, each counter group in its own run), FETCH/WRITE_SIZE calibration probe.
# Usage on the GPU box: bash tools/profile_round.sh <tag>   (outputs under gpurun_out/<tag>/)
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python bench.py > $O/bench.json 2> $O/bench.err )
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-concurrent > $O/stats_bench.json 2> $O/stats.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1 -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq2 -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
if [ -x $R/tools/probe/fetch_calib ]; then
  timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- $R/tools/probe/fetch_calib > $O/calib.log 2>&1
  timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- $R/tools/probe/fetch_calib >> $O/calib.log 2>&1
  python - <<PY
import csv, glob, collections
for d, c in (('$O/calib_fetch', 'FETCH_SIZE'), ('$O/calib_write', 'WRITE_SIZE')):
    rows = collections.defaultdict(list)
    for p in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            if r['Counter_Name'] == c: rows[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    for k, v in rows.items(): print('calib', c, k, 'mean KB', sum(v) / len(v))
PY
fi
cat $O/bench.json; tail -2 $O/bench.err
python $R/tools/pmc_summary.py $O/pmc_fetch; python $R/tools/pmc_summary.py $O/pmc_write
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -15
# round 3: host-side numbers of the drop-in surface, GraphedRenderer frame times, the five-render iteration per kernel
( cd $R && timeout 300 python tools/gpu_host_profile.py > $O/host_profile.log 2>&1; timeout 300 python tools/gpu_graphed_times.py > $O/graphed.log 2>&1 )
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iter_sets -- python $R/tools/gpu_iteration_profile.py sets 40 > $O/iter_sets.log 2>&1
# round 4: the graphed five-render iteration per kernel; C5 (configs[4]) per kernel; host / device split of GraphedIteration
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iter_graphed -- python $R/tools/gpu_iteration_profile.py graphed 60 > $O/iter_graphed.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- python $R/bench.py --config c5 --steps 100 --warmup 10 --no-cpu-baseline --no-concurrent --no-other-configs > $O/c5_bench.json 2> $O/c5.err
( cd $R && timeout 300 python tools/gpu_graphed_iter_profile.py > $O/graphed_iter_profile.log 2>&1 )
cat $O/graphed_iter_profile.log | grep -v amdgpu
grep "eager render\|^host:" $O/host_profile.log; cat $O/graphed.log | cut -c1-160
# round 5: how every kernel scales with K identical views per launch (latency- vs throughput-bound)
( cd $R && timeout 200 python tools/gpu_kernel_times_k.py 0 2>&1 | grep '^K=' | tee $O/kernel_times_k.log )
# round 4 (late): the same iteration with the loss recorded into the graph (timing only)
( cd $R && for how in graphed graphed_loss sets; do timeout 200 python tools/gpu_iteration_profile.py $how 300 2>&1 | tail -1; done ) | tee $O/iter_times.log
