#!/bin/bash
# round 3: LDS-DMA staging of the grouped GEMM (SET_GEMM_DMA=1) against the register-staged kernel, one GPU box:
# parity tests with the switch on, the C++ microbenchmark of the step's three launch shapes, and a bench A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m show_edit_tell_amd.build > /dev/null 2>&1
SET_GEMM_DMA=1 timeout 600 python -m pytest tests/test_hip_shapes.py tests/test_hip_editnet.py tests/test_hip_ops.py -m gpu -x -q 2>&1 | tail -5
bash tools/ubench/build_gemm_steps.sh r3 2>&1 | tail -3
for v in 0 1 0 1; do echo "== SET_GEMM_DMA=$v"; SET_GEMM_DMA=$v timeout 300 tools/ubench/gemm_steps_r3 1000 2>&1 | grep -v "^     " ; done
AB_STEPS=100 bash tools/ab_env.sh "SET_GEMM_DMA=0" "SET_GEMM_DMA=1"
