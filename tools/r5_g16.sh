#!/bin/bash
# merged backward launches + grouped colsum: tests, train timing, kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5
mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_sequence.py tests/test_hip_train.py tests/test_hip_loss.py -x -q -m gpu > $O/t16_tests.log 2>&1
tail -5 $O/t16_tests.log
for m in 1; do echo -n "merged=$m "; SET_XE_BWD_MERGED=$m python tools/bench_train.py --steps 10 --warmup 3 2>&1 | grep -o "ms_per_train_step.: [0-9.]*"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_train16 -o train -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 7 --warmup 0 > $GRAFT_REPO_ROOT/$O/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $O/prof_train16/train_results.db > $O/t16_train_kernel_stats.txt
rm -rf $O/prof_train16
head -50 $O/t16_train_kernel_stats.txt
