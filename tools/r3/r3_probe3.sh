#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
python -m pytest tests/test_hip_atsize.py tests/test_hip_dp.py -m gpu -q -x 2>&1 | tail -8
for b in 4 16 32; do python tools/profile_small_batch.py $b 2>&1 | grep -v amdgpu.ids; done
SET_PROFILE_SITES=1 python tools/profile_small_batch.py 4 2>&1 | grep "gemm:"
