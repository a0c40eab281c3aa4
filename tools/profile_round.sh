#!/bin/bash
# ONE lease, ONE build: everything the committed summaries of a round are made from.  Writes gpurun_out/<tag>/ on the GPU box:
#   digest.txt          digest of the library sources this lease ran (exavatar_release_amd.build._digest) -- every summary is stamped
#                       with it and bench.py quotes a summary only when it matches the loaded library
#   pytest_gpu.log      the GPU suite (its parity statistics land in gpurun_out/parity_stats.jsonl, started empty here)
#   bench.json          the default bench line
#   stats/              rocprofv3 --kernel-trace --stats of the bench command
#   pmc_*/ calib_*/     HBM-traffic and SQ counter passes (counters only with --kernel-trace, each group in its own run) + the
#                       FETCH/WRITE_SIZE calibration probe
#   c5_stats/, iter_*   C5 per kernel, the five-render iteration per kernel, K-scaling, host time of the autograd surface
# Usage on the GPU box: bash tools/profile_round.sh <tag>;  then, back home: python tools/make_profiles.py gpurun_out/<tag> <tag>
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && python -c "from exavatar_release_amd import build as b; print(b._digest()[:16])" > $O/digest.txt )
rm -f $R/gpurun_out/parity_stats.jsonl
( cd $R && timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log; cp $R/gpurun_out/parity_stats.jsonl $O/ 2>/dev/null; cp $R/gpurun_out/compiled_node_calls.txt $O/ 2>/dev/null )
( cd $R && timeout 900 python bench.py > $O/bench.json 2> $O/bench.err )
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-concurrent > $O/stats_bench.json 2> $O/stats.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1 -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq2 -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
if [ ! -x $R/tools/probe/fetch_calib ]; then ( cd $R/tools/probe && hipcc -w --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ) 2>/dev/null; fi
if [ -x $R/tools/probe/fetch_calib ]; then
  timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- $R/tools/probe/fetch_calib > $O/calib.log 2>&1
  timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- $R/tools/probe/fetch_calib >> $O/calib.log 2>&1
fi
cat $O/bench.json | cut -c1-400; tail -2 $O/bench.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -15
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- python $R/bench.py --config c5 --steps 100 --warmup 10 --no-cpu-baseline --no-concurrent --no-other-configs > $O/c5_bench.json 2> $O/c5.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iter_graphed -- python $R/tools/gpu_iteration_profile.py graphed 60 > $O/iter_graphed.log 2>&1
( cd $R && timeout 200 python tools/gpu_kernel_times_k.py 0 2>&1 | grep '^K=' | tee $O/kernel_times_k.log )
( cd $R && timeout 200 python tools/gpu_surface_host.py 300 2>&1 | grep -v amdgpu | tee $O/surface_host.log; timeout 100 python tools/gpu_surface_host_small.py 2>&1 | grep -v amdgpu | head -6 | tee -a $O/surface_host.log )
( cd $R && for how in graphed graphed_loss sets; do timeout 200 python tools/gpu_iteration_profile.py $how 300 2>&1 | tail -1; done ) | tee $O/iter_times.log
# trim what travels back (<= 64 MiB): the traces themselves are not needed, the per-kernel csv summaries are
find $O -name "*_kernel_trace.csv" -size +2M -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
du -sh $O
