cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r5/t15_tests.log
tail -4 gpurun_out/r5/t15_tests.log
for v in 1 0; do echo -n "slab=$v "; SET_SLAB_DIRECT=$v python tools/bench_train.py --steps 10 --warmup 3 2>&1 | grep -o "ms_per_train_step.: [0-9.]*"; done
python tools/bench_scst.py 2>&1 | grep -o "ms_per_step.: [0-9.]*"
