#!/bin/bash
# PMC comparison of the fp32 kernel and the split-precision kernel on a steady-state shape (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SET_GEMM_TARGET_WGS=1 ITERS=3
SH="128 32768 4096"
for MODE in 0 2; do
  OUT=gpurun_out/pmc_split/m$MODE
  mkdir -p $OUT
  export SET_GEMM_SPLIT=$MODE
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/p1 -o p1 -- python tools/gemm_microbench.py $SH > $OUT/p1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p2 -o p2 -- python tools/gemm_microbench.py $SH > $OUT/p2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p3 -o p3 -- python tools/gemm_microbench.py $SH > $OUT/p3.log 2>&1
  for p in p1 p2 p3; do python tools/pmc_table.py $(dirname $(ls $OUT/$p/*/*_counter_collection.csv $OUT/$p/*_counter_collection.csv 2>/dev/null | head -1)) $p 2>/dev/null | grep -A12 "gemm_nt" ; done > $OUT/summary.txt
  tail -2 $OUT/p1.log
done
