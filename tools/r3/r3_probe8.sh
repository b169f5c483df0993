#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
timeout 1200 python -m pytest tests/test_hip_train.py tests/test_hip_train_mode.py tests/test_hip_sequence.py tests/test_hip_dp.py tests/test_hip_atsize.py -m gpu -q -x 2>&1 | tail -6
for v in 0 1 0 1; do echo "SET_WGRAD_OVERLAP=$v"; SET_WGRAD_OVERLAP=$v python tools/bench_train.py --steps 12 --warmup 4 2>&1 | grep -v amdgpu.ids | tail -2; done
