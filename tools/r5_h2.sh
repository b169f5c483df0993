#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_editnet.py tests/test_hip_dcnet.py tests/test_hip_ops.py tests/test_hip_sequence.py tests/test_hip_train.py tests/test_hip_beam.py -x -q -m gpu 2>&1 | tail -3
AB_STEPS=100 bash tools/ab_env.sh "SET_ENC_UNITS16=0" 2>&1 | cut -c1-300
