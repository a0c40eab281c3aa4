#!/bin/bash
# Final check of a round, part B: the whole GPU suite (parity statistics land in gpurun_out/parity_stats.jsonl).
TAG=${1:-r02z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
rm -f gpurun_out/parity_stats.jsonl
timeout ${2:-300} python -m pytest tests -m gpu -q --timeout 150 --durations=8 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/summary.log
grep -E "^E  |FAILED|passed|failed|s call" $O/pytest.log | cut -c1-220 | head -40
