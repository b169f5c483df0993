#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_sequence.py tests/test_hip_train.py tests/test_hip_train_mode.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do python tools/bench_train.py --steps 10 --warmup 3 2>&1 | grep -o "ms_per_train_step.: [0-9.]*"; done
python tools/train_torch_sites.py 2>/dev/null | head -12
