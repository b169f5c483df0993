#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
python tools/bench_scst.py --steps 6 --warmup 2 2>&1 | tail -1 > $O/t18_scst.json; grep -o "ms_per_step.: [0-9.]*" $O/t18_scst.json
python tools/scst_breakdown.py > $O/t18_scst_breakdown.txt 2>&1; tail -12 $O/t18_scst_breakdown.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_scst -o scst -- python $GRAFT_REPO_ROOT/tools/bench_scst.py --steps 4 --warmup 0 > $GRAFT_REPO_ROOT/$O/prof_scst.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $O/prof_scst/scst_results.db > $O/t18_scst_kernel_stats.txt
rm -rf $O/prof_scst
head -45 $O/t18_scst_kernel_stats.txt
