#!/bin/bash
# round-4 GEMM probe on the GPU box: parity of gemm_nt_f32_wreg against gemm_nt_f32<64,64> + timings of the step launches
mkdir -p gpurun_out/r4
B=tools/ubench/gemm_steps_r4
IT=${IT:-1500}
{
echo "== check"; CHECK_WREG=1 $B 50 | grep -E "^check|^F/A:"
echo "== base"; $B $IT | head -12
echo "== base ZERO_INPUT"; ZERO_INPUT=1 $B $IT | head -6
echo "== SET_GEMM_WREG=1"; SET_GEMM_WREG=1 $B $IT | head -12
echo "== SET_GEMM_WREG=1 ZERO_INPUT"; ZERO_INPUT=1 SET_GEMM_WREG=1 $B $IT | head -6
echo "== SET_GEMM_WREG=1 SET_GEMM_WGS64_PCT=100"; SET_GEMM_WREG=1 SET_GEMM_WGS64_PCT=100 $B $IT | head -6
} > gpurun_out/r4/gemm_probe_${TAG:-a}.txt 2>&1
