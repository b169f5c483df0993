#!/bin/bash
cd $GRAFT_REPO_ROOT
SET_FUSED_NW8=4 timeout 2000 python -m pytest tests/test_hip_editnet.py tests/test_hip_ops.py tests/test_hip_dcnet.py -x -q -m gpu 2>&1 | tail -3
SET_FUSED_NW8=5 timeout 2000 python -m pytest tests/test_hip_editnet.py -x -q -m gpu 2>&1 | tail -2
AB_STEPS=100 bash tools/ab_env.sh "SET_FUSED_NW8=3" "SET_FUSED_NW8=4" "SET_FUSED_NW8=5" 2>&1 | cut -c1-300
