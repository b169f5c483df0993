cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
for v in 0 1; do
SET_SLAB_DIRECT=$v rocprofv3 --kernel-trace --stats -d $O/prof_train$v -o train -- python tools/bench_train.py --steps 7 --warmup 0 > $O/prof_train.log 2>&1
python tools/rocprof_summary.py $O/prof_train$v/train_results.db > $O/t14_train_kernel_stats_slab$v.txt
rm -rf $O/prof_train$v
echo "== SET_SLAB_DIRECT=$v"; head -14 $O/t14_train_kernel_stats_slab$v.txt | cut -c1-140; grep -E "slab_reduce|dropout_bwd_philox|lstm_cell_bwd|copy_gate_bwd|select_bwd|context_gate_bwd|attention_bwd|lstm_gates_bwd" $O/t14_train_kernel_stats_slab$v.txt | cut -c1-140
done
