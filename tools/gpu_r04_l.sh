#!/bin/bash
# Round-4 check L: GraphedIteration(loss_fn=...) -- forward + loss + backward in one hipGraph.
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_graphed_iteration.py tests/test_gpu_fold.py -x -q 2>&1 | tail -15
for i in 1 2; do
  for how in graphed graphed_loss sets; do
    timeout 200 python tools/gpu_iteration_profile.py $how 300 2>&1 | tail -1
  done
done
