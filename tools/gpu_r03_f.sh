#!/bin/bash
# Round-3 check F: fused scatter + sort launch (binsort_kernel): GPU suite, phase probe, A/B of the kernel chain with the knob.
TAG=${1:-r03f}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.log; tail -5 $O/pytest.log | cut -c1-300
[ -f exavatar_release_amd/_variants/bsprobe.so ] && EXA_RASTER_LIB=exavatar_release_amd/_variants/bsprobe.so timeout 200 python tools/gpu_binsort_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-250
ab() {
  echo "== $*"
  env "$@" timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-concurrent --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['roofline']['kernel_avg_us'].items()})"
}
ab EXA_FUSED_BINSORT=1
ab EXA_FUSED_BINSORT=0
ab EXA_FUSED_BINSORT=1
ab EXA_FUSED_BINSORT=0
