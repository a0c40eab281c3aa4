#!/bin/bash
# Round-4 check R: the tight backward recording when the host reaches the backward before the composites' reports (bounded wait).
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_graphed_iteration.py -x -q 2>&1 | tail -2
python tools/gpu_iter_repeat.py 2>&1 | tail -2 | cut -c1-420
