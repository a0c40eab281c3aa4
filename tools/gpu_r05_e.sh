#!/bin/bash
# Round 5, run E: 64-byte partial records -- suite, A/B against the 48-byte layout of the same tree.
mkdir -p gpurun_out/r05e
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_bench_ranks.py > gpurun_out/r05e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05e/pytest.log
tail -4 gpurun_out/r05e/pytest.log
bash tools/gpu_ab.sh -n 3 exavatar_release_amd/_variants/p48.so exavatar_release_amd/libexa_raster.so | tee gpurun_out/r05e/ab.log
