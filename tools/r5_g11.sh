cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof_train -o train -- python tools/bench_train.py --steps 7 --warmup 0 > $O/prof_train.log 2>&1
python tools/rocprof_summary.py $O/prof_train/train_results.db > $O/t11_train_kernel_stats.txt
rm -rf $O/prof_train
head -45 $O/t11_train_kernel_stats.txt | cut -c1-150
