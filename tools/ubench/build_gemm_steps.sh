#!/bin/bash
# build tools/ubench/gemm_steps_<tag> (self-contained, includes csrc/gemm_f32.hip):  build_gemm_steps.sh [tag] [extra hipcc flags...]
cd "$(dirname "$0")"
tag=${1:-base}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-comment -Wno-unused-result ${PRELOAD--mllvm -amdgpu-kernarg-preload-count=8} "$@" gemm_steps.hip -o gemm_steps_$tag
