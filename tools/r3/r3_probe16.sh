#!/bin/bash
# in-launch split-K combine for the training GEMMs (SET_GEN_COMBINE=1): gradient parity, then A/B of the XE step
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
SET_GEN_COMBINE=1 timeout 1200 python -m pytest tests/test_hip_train.py tests/test_hip_train_mode.py tests/test_hip_sequence.py tests/test_hip_ops.py -m gpu -q -x 2>&1 | tail -3
for v in 0 1 0 1; do
  echo "== SET_GEN_COMBINE=$v"
  SET_GEN_COMBINE=$v python tools/bench_train.py --steps 30 2>&1 | grep -v amdgpu.ids | tail -2
done
