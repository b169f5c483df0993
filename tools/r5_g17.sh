#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for h in 1 0; do echo -n "step_logs=$h "; SET_XE_STEP_LOGS=$h python tools/bench_train.py --steps 10 --warmup 3 2>&1 | grep -o "ms_per_train_step.: [0-9.]*"; done
done
timeout 1200 python -m pytest tests/test_hip_sequence.py tests/test_hip_train.py -x -q -m gpu 2>&1 | tail -3
