#!/bin/bash
# Round-4 check S: inside captured graphs the composites' list merges run on a side stream while their sources blend.
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_graphed_iteration.py tests/test_gpu_soak.py tests/test_gpu_fold.py tests/test_gpu_cabi.py -x -q 2>&1 | tail -6 | cut -c1-250
for i in 1 2 3; do
  for o in 1 0; do echo -n "EXA_OVERLAP=$o "; EXA_OVERLAP=$o timeout 200 python tools/gpu_iteration_profile.py graphed 300 2>&1 | tail -1; done
done
