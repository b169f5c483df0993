#!/bin/bash
# where a k-tile of the experimental bf16-split GEMM spends its time: diagnostic builds of the experimental library variant
# (results are garbage in them), microbenchmark of the F/A-like and the D launch shapes.  tools/split_diag.sh
cd $GRAFT_REPO_ROOT
export SET_LIB_VARIANT=exp SET_GEMM_SPLIT=1 ITERS=100
for f in "" "-DSPL_EXP_NOSPLIT" "-DSPL_EXP_NOMFMA" "-DSPL_EXP_NOGLOAD" "-DSPL_EXP_NOSPLIT -DSPL_EXP_NOGLOAD" "-DSPL_EXP_NOSPLIT -DSPL_EXP_NOMFMA -DSPL_EXP_NOGLOAD"; do
  SET_HIPCC_FLAGS="$f" python -m show_edit_tell_amd.build --exp --force > /dev/null 2>&1
  echo "== [$f]"
  python tools/gemm_microbench.py 128 18192 1024 128 4096 3072 2>&1 | grep -v amdgpu.ids
done
python -m show_edit_tell_amd.build --exp --force > /dev/null 2>&1
