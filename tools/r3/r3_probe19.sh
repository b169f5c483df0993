#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
timeout 600 python -m pytest tests/test_hip_dp.py -m gpu -q -x -k "rccl_with_one_rank" 2>&1 | tail -40
