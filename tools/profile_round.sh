#!/bin/bash
# Round profile: default bench line, rocprofv3 kernel stats of the bench command, HBM-traffic and SQ PMC passes.
# Usage on the GPU box: bash tools/profile_round.sh <tag>   (outputs under gpurun_out/<tag>/)
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 python bench.py > $O/bench.json 2> $O/bench.err )
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-concurrent > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1 -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq2 -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
cat $O/bench.json; tail -2 $O/bench.err
python $R/tools/pmc_summary.py $O/pmc_fetch; python $R/tools/pmc_summary.py $O/pmc_write
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -15
