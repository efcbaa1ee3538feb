cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/r02_counters_list.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmc_r02_sq1 -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --plan 16,160,32 > gpurun_out/pmc_r02_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/pmc_r02_sq2 -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --plan 16,160,32 > gpurun_out/pmc_r02_sq2.log 2>&1
