#!/bin/bash
# One parameterised GPU-box probe instead of a script per experiment (the 21 one-off r3_probe*.sh of round 3 were exactly
# these lines with different switches):
#   tools/probe.sh tests [pytest args]          GPU test suite (default: tests -m gpu -q -x)
#   tools/probe.sh ab "ENV=.. ENV=.." "..."     bench A/B of runtime switches, ABAB order (tools/ab_env.sh)
#   tools/probe.sh bench [bench.py args]        one bench line, JSON to gpurun_out/probe_bench.json
#   tools/probe.sh gemm [iters]                 GEMM step microbenchmark (tools/ubench), default kernel + SET_* from the env
#   tools/probe.sh small [B ...]                per-kernel profile of a small-batch decode (tools/profile_small_batch.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m show_edit_tell_amd.build > /dev/null 2>&1
what=${1:-tests}; shift
case $what in
  tests) timeout 1700 python -m pytest ${@:-tests -m gpu -q -x} 2>&1 | tail -8 ;;
  ab)    bash tools/ab_env.sh "$@" ;;
  bench) python bench.py "$@" > gpurun_out/probe_bench.json 2> gpurun_out/probe_bench.err; tail -c 2000 gpurun_out/probe_bench.json ;;
  gemm)  bash tools/ubench/build_gemm_steps.sh probe > /dev/null 2>&1; tools/ubench/gemm_steps_probe ${1:-1500} | grep -v "^     " ;;
  small) for b in ${@:-4}; do python tools/profile_small_batch.py $b 2>&1 | grep -v amdgpu.ids; done ;;
  *) echo "unknown probe $what"; exit 2 ;;
esac
