#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_gemm2
mkdir -p $OUT
export SET_GEMM_TARGET_WGS=1 ITERS=3
SH="128 32768 2048 128 32768 8192"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/p1 -o p1 -- python tools/gemm_microbench.py $SH > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p2 -o p2 -- python tools/gemm_microbench.py $SH > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p3 -o p3 -- python tools/gemm_microbench.py $SH > $OUT/p3.log 2>&1
tail -3 $OUT/p3.log
