cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_hip_train.py tests/test_hip_train_mode.py tests/test_hip_sequence.py tests/test_hip_sampling.py tests/test_hip_atsize.py tests/test_hip_dp.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r5/t12_tests.log
tail -6 gpurun_out/r5/t12_tests.log
for v in 0 1; do SET_SLAB_DIRECT=$v timeout 300 python tools/bench_train.py --steps 10 --warmup 3 2>&1 | grep ms_per | cut -c1-150; done
