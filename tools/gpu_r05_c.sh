#!/bin/bash
# Round 5, run C: per-strip lists in render_fwd -- suite, stress, A/B against the round-4 kernels.
mkdir -p gpurun_out/r05c
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r05c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05c/pytest.log
tail -4 gpurun_out/r05c/pytest.log
PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 600 python tools/gpu_overflow_stress.py 1 > gpurun_out/r05c/stress.log 2>&1; echo "stress rc=$?" >> gpurun_out/r05c/stress.log
timeout 600 python tools/gpu_overflow_stress.py 1 >> gpurun_out/r05c/stress.log 2>&1; echo "stress (caching) rc=$?" >> gpurun_out/r05c/stress.log
grep "rc=\|STRESS" gpurun_out/r05c/stress.log
bash tools/gpu_ab.sh -n 2 exavatar_release_amd/_variants/r04.so exavatar_release_amd/libexa_raster.so | tee gpurun_out/r05c/ab.log
