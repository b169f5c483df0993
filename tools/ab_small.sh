#!/bin/bash
# A/B of RUNTIME knobs at a small / mid batch on ONE GPU box: tools/ab_small.sh "<B ...>" "<ENV=.. ENV=..>" "<...>" ...
# (per-kernel HIP-event breakdown of tools/profile_small_batch.py; first line of each = ms per decode)
cd $GRAFT_REPO_ROOT
bs=$1; shift
for cfg in "$@"; do
  for b in $bs; do
    echo "== [$cfg] B=$b"
    env $cfg python tools/profile_small_batch.py $b 2>&1 | grep -v amdgpu.ids | head -${AB_LINES:-9}
  done
done
