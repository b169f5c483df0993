#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
for b in 4 128; do timeout 300 python tools/profile_small_batch.py $b 2>&1 | grep -v amdgpu.ids | head -7; done
AB_STEPS=100 bash tools/ab_env.sh "SET_ATT_V2=0" "SET_ATT_V2=1"
