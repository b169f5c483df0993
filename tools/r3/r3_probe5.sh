#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
timeout 1200 python -m pytest tests/test_hip_editnet.py tests/test_hip_ops.py tests/test_hip_dcnet.py tests/test_hip_boundary.py tests/test_hip_atsize.py tests/test_hip_shapes.py tests/test_hip_beam.py -m gpu -q -x 2>&1 | tail -8
for b in 4 16 32; do timeout 300 python tools/profile_small_batch.py $b 2>&1 | grep -v amdgpu.ids | head -9; done
echo "== B=4 fused E"; SET_FUSED_MIN_ROWS=1 timeout 300 python tools/profile_small_batch.py 4 2>&1 | grep -v amdgpu.ids | head -9
echo "== B=4 old paths"; SET_GEMM_BM16_UPTO=0 SET_ATT_SMALL_SLICES=0 SET_ENC_PERSISTENT=0 timeout 300 python tools/profile_small_batch.py 4 2>&1 | grep -v amdgpu.ids | head -3
echo "== B=128 persistent"; SET_ENC_PERSISTENT_MAXB=128 timeout 300 python tools/profile_small_batch.py 128 2>&1 | grep -v amdgpu.ids | grep "ms per\|persistent"
SET_PROFILE_SITES=1 python tools/profile_small_batch.py 4 2>&1 | grep "gemm:"
