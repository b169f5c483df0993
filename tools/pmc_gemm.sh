#!/bin/bash
# PMC passes over the GEMM microbenchmark (counters only with --kernel-trace; separate passes).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_gemm
mkdir -p $OUT
SH="${SHAPES:-2560 4096 1024 128 4096 3072}"
ITERS=5 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/p1 -o p1 -- python tools/gemm_microbench.py $SH > $OUT/p1.log 2>&1
ITERS=5 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p2 -o p2 -- python tools/gemm_microbench.py $SH > $OUT/p2.log 2>&1
ITERS=5 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -o p3 -- python tools/gemm_microbench.py $SH > $OUT/p3.log 2>&1
ITERS=5 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -o p4 -- python tools/gemm_microbench.py $SH > $OUT/p4.log 2>&1
find $OUT -name "*.csv" | head
