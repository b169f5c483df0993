#!/bin/bash
cd $GRAFT_REPO_ROOT
AB_STEPS=100 bash tools/ab.sh "" "-DSET_EXP_VGPR_CAP_ATT=3" "-DSET_EXP_VGPR_CAP_ATT=4" 2>&1 | cut -c1-260
