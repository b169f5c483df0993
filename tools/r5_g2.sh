set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_finished_rows.py tests/test_hip_train.py tests/test_hip_sequence.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5/t2_tests.log
for hwb in 0 1; do
  SET_GEMM_GEN_HWB=$hwb timeout 300 python tools/bench_wgrad_shapes.py > gpurun_out/r5/t2_wgrad_hwb$hwb.txt 2>&1
  SET_GEMM_GEN_HWB=$hwb timeout 300 python tools/bench_train.py --steps 10 --warmup 3 > gpurun_out/r5/t2_train_hwb$hwb.json 2>&1
done
tail -3 gpurun_out/r5/t2_tests.log
cat gpurun_out/r5/t2_train_hwb*.json | grep ms_per
