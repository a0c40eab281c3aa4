#!/bin/bash
# Round 5, run A: (1) overflow stress against the round-4 library (exavatar_release_amd/_variants/r04.so = HEAD of round 4) and
# against the fixed one, every workspace its own hipMalloc; (2) the GPU suite, uncaptured, log kept.
mkdir -p gpurun_out/r05a
export AMD_LOG_LEVEL=1
for lib in exavatar_release_amd/_variants/r04.so exavatar_release_amd/libexa_raster.so; do
  for i in 1 2 3; do
    echo "=== stress $lib run $i" >> gpurun_out/r05a/stress.log
    EXA_RASTER_LIB=$PWD/$lib PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 600 python tools/gpu_overflow_stress.py 2 >> gpurun_out/r05a/stress.log 2>&1
    echo "rc=$?" >> gpurun_out/r05a/stress.log
  done
done
unset AMD_LOG_LEVEL
timeout 1500 python -m pytest tests/ -x -q -m gpu -s -p no:cacheprovider > gpurun_out/r05a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05a/pytest.log
tail -5 gpurun_out/r05a/pytest.log
grep -E "^===|^rc=|STRESS_OK|fault|Fault|abort" gpurun_out/r05a/stress.log | head -60
