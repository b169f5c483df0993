#!/bin/bash
cd $GRAFT_REPO_ROOT
export SET_ENC_UNITS16=0
AB_STEPS=60 bash tools/ab.sh "-DSET_EXP_ENC_NOK=1" "-DSET_EXP_ENC_NOK=4" "" 2>&1 | cut -c1-150 | head -3
