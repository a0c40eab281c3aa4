cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc1 -- python $R/tools/gpu_kernel_times.py 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc2 -- python $R/tools/gpu_kernel_times.py 0 > $R/gpurun_out/pmc2.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc2
tail -3 $R/gpurun_out/pmc2.log
