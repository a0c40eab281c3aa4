#!/bin/bash
# Round-4 check A: knob tests + full GPU suite on the working build; A/B of the render_bwd LDS fix against the round-3
# library (EXA_RASTER_LIB); SQ PMC pass (LDS counters) of the working build.
TAG=${1:-r04a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_knobs.py -q --timeout 250 > $O/pytest_knobs.log 2>&1
echo "knobs rc=$?" | tee -a $O/summary.log; tail -25 $O/pytest_knobs.log | cut -c1-300
timeout 600 python -m pytest tests -m gpu -q --timeout 300 --deselect tests/test_gpu_knobs.py > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.log; tail -8 $O/pytest.log | cut -c1-300
ab() {
  echo "== $1"
  env $1 timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
}
ab EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/r03base.so
ab EXA_X=0
ab EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/r03base.so
ab EXA_X=0
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq2 -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1 -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq1
