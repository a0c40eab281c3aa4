#!/bin/bash
# Round-4 check P: backward launches bounded by the batch slots in use (used_slots): tests, then the iteration tight / full.
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_graphed_iteration.py tests/test_gpu_fold.py tests/test_gpu_soak.py tests/test_gpu_cabi.py -x -q 2>&1 | tail -8 | cut -c1-250
for i in 1 2; do
  for t in 1 0; do echo -n "EXA_TIGHT=$t "; EXA_TIGHT=$t timeout 200 python tools/gpu_iteration_profile.py graphed 300 2>&1 | tail -1; done
  timeout 200 python tools/gpu_iteration_profile.py sets 300 2>&1 | tail -1
done
