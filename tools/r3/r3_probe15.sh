#!/bin/bash
# 128x128 tiles for the LARGE products (training: time-batched contractions): potential check
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
bash tools/ubench/build_gemm_steps.sh r3 2>&1 | grep -E "error" | head
for v in 0 1; do
  echo "== SET_GEMM_BN128=$v"
  SET_GEMM_BN128=$v timeout 120 tools/ubench/gemm_steps_r3 2000 2>&1 | grep -E "big"
  SET_GEMM_BN128=$v timeout 300 python tools/bench_wgrad_shapes.py 2>&1 | grep -E "fwd|dgrad|wgrad x2h|wgrad fc"
done
