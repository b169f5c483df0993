#!/bin/bash
# stability soak: the GPU suite three times, then the grid-barrier / concurrency tests twenty times
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
for i in 1 2 3; do
  timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1
done
for i in $(seq 1 20); do
  timeout 300 python -m pytest tests/test_hip_boundary.py -m gpu -q -x -p no:cacheprovider -k "persistent or concurrent or stream" 2>&1 | tail -1
done
python __graft_entry__.py smoke 2>&1 | tail -1
