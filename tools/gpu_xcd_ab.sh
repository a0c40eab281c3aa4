#!/bin/bash
# A/B of launch-order variants (tools/build_variant.sh WORK <name> -DEXA_XCD_REGIONS=1 ...): step times (tools/gpu_ab.sh) and the
# HBM-side traffic of the blends (FETCH_SIZE / WRITE_SIZE, one counter per pass, C3 ring views 0 / 50 / 100).
# Usage on the GPU box: bash tools/gpu_xcd_ab.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
bash tools/gpu_ab.sh -n 2 "$@"
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/xcd; rm -rf $O; mkdir -p $O
for lib in "$@"; do
  v=$(basename $lib .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    EXA_RASTER_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${v}_$c -- python $GRAFT_REPO_ROOT/tools/gpu_kernel_times.py 0 50 100 > /dev/null 2>&1
    echo "== $v $c (KB per launch)"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $O/${v}_$c | grep -E "render_|preprocess_bwd"
  done
done
rm -rf $O
