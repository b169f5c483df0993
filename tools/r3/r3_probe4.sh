#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
timeout 900 python -m pytest tests/test_hip_editnet.py tests/test_hip_ops.py tests/test_hip_dcnet.py tests/test_hip_boundary.py tests/test_hip_atsize.py -m gpu -q -x 2>&1 | tail -8
for b in 4 128; do timeout 300 python tools/profile_small_batch.py $b 2>&1 | grep -v amdgpu.ids | head -8; done
AB_STEPS=100 bash tools/ab_env.sh "SET_ENC_PERSISTENT=0" "SET_ENC_PERSISTENT=1"
