#!/bin/bash
# SQ / TCC counter passes of the bench command (separate runs, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace -d gpurun_out/pmc_sq2 -o sq --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --pipeline 0 > gpurun_out/pmc_sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --kernel-trace -d gpurun_out/pmc_tcc2 -o tcc --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --pipeline 0 > gpurun_out/pmc_tcc2.log 2>&1
ls gpurun_out/pmc_sq2 gpurun_out/pmc_tcc2
