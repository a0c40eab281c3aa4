#!/bin/bash
# Round 5, run F: the prefetched view switch (forked graph branch) against the round-4 protocol; multi-rank bench tests.
mkdir -p gpurun_out/r05f
for i in 1 2; do for m in prefetch graph; do
  EXA_BENCH_CAM_COPY=$m timeout 250 python bench.py --no-cpu-baseline --no-concurrent --no-other-configs --no-kernel-timing 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$m', round(d['value'], 1), round(d['ms_per_step'], 4), d['config']['view_switch'][:60])" | tee -a gpurun_out/r05f/ab.log
done; done
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r05f/ranks.log
