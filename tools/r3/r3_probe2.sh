#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
python tools/r3/r3_split_probe.py 2>&1 | grep -v amdgpu.ids
bash tools/ubench/build_gemm_steps.sh r3 2>&1 | grep -c error
for v in 0 1; do echo "== SET_GEMM_DMA=$v"; SET_GEMM_DMA=$v timeout 300 tools/ubench/gemm_steps_r3 1000 2>&1 | grep -v "^     " ; done
