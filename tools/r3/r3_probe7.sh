#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
SET_GEMM_BN32=1 timeout 600 python -m pytest tests/test_hip_editnet.py tests/test_hip_shapes.py -m gpu -x -q 2>&1 | tail -3
AB_STEPS=100 bash tools/ab_env.sh "SET_GEMM_BN32=0" "SET_GEMM_BN32=1"
PMC_NO_SQ=1 PMC_STREAMS=1 bash tools/pmc_bench.sh gpurun_out/pmc_s1_base > /dev/null 2>&1; head -c 1500 gpurun_out/pmc_s1_base/traffic.json; echo
PMC_NO_SQ=1 PMC_STREAMS=1 SET_GEMM_BN32=1 bash tools/pmc_bench.sh gpurun_out/pmc_s1_bn32 "gemm_nt_f32<128, 32" > /dev/null 2>&1; head -c 1500 gpurun_out/pmc_s1_bn32/traffic.json; echo
PMC_NO_SQ=1 PMC_STREAMS=3 bash tools/pmc_bench.sh gpurun_out/pmc_s3_base > /dev/null 2>&1; head -c 1500 gpurun_out/pmc_s3_base/traffic.json; echo
rm -rf gpurun_out/pmc_s*/fetch gpurun_out/pmc_s*/write
