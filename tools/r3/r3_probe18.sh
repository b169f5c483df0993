#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
timeout 900 python -m pytest tests/test_hip_shapes.py -m gpu -q -x 2>&1 | tail -15
