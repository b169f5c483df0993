#!/bin/bash
cd $GRAFT_REPO_ROOT
AB_STEPS=100 bash tools/ab.sh "" "-DSET_EXP_VGPR_CAP=4" "-DSET_EXP_VGPR_CAP=3" 2>&1 | cut -c1-260
