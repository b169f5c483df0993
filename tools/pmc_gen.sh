#!/bin/bash
# PMC passes over the training step's large products (NT forward, NN dX, TN dW): what separates the general-layout kernel's
# 110 TFLOP/s from the forward kernel's 124-127
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
OUT=gpurun_out/pmc_gen
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/p1 -o p1 -- python tools/bench_wgrad_shapes.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $OUT/p2 -o p2 -- python tools/bench_wgrad_shapes.py > $OUT/p2.log 2>&1
for p in p1 p2; do
  d=$(dirname $(find $OUT/$p -name "${p}_counter_collection.csv" | head -1))
  python tools/pmc_table.py $d $p | grep -A12 "gemm_gen_f32\|gemm_nt_f32" > $OUT/$p.txt
done
rm -rf $OUT/p1 $OUT/p2
tail -20 $OUT/p1.log
