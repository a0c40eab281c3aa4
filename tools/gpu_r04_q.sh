#!/bin/bash
# Round-4 check Q: the batch-major launch order of the backward for batches of up to three jobs (the iteration's plain renders).
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do
  echo -n "default   "; timeout 200 python tools/gpu_iteration_profile.py graphed 300 2>&1 | tail -1
  echo -n "order K<=3 "; EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/order_k3.so timeout 200 python tools/gpu_iteration_profile.py graphed 300 2>&1 | tail -1
done
EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/order_k3.so timeout 600 python -m pytest tests/test_gpu_graphed_iteration.py tests/test_gpu_fold.py -x -q 2>&1 | tail -3 | cut -c1-200
