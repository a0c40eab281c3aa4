#!/bin/bash
# Round-4 check K: is_vis from the forward kernel, lazy composite radius / is_vis, composite gradients folded into the source's
# per-Gaussian backward kernel.  New tests + the iteration tests, then the iteration's time with the fold on / off.
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_fold.py tests/test_gpu_graphed_iteration.py tests/test_gpu_soak.py tests/test_gpu_knobs.py -x -q 2>&1 | tail -15
for i in 1 2; do
  for f in 1 0; do
    for how in graphed sets; do
      echo -n "EXA_FOLD=$f "; EXA_FOLD=$f timeout 200 python tools/gpu_iteration_profile.py $how 300 2>&1 | tail -1
    done
  done
done
