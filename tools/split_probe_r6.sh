#!/bin/bash
# Round 6: the experimental bf16-split emulated-fp32 GEMM (csrc/experimental/gemm_variants.inc) on the experimental library
# variant (python -m show_edit_tell_amd.build --exp): fp32 kernel vs SET_GEMM_SPLIT=1 (both operands split on the fly, round 1)
# vs SET_GEMM_SPLIT=2 (pre-split weight planes).  Microbenchmark (time + max error vs fp64), golden parity tests, bench A/B.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
export SET_LIB_VARIANT=exp
for m in 0 1 2 3; do
  echo "== SET_GEMM_SPLIT=$m"
  SET_GEMM_SPLIT=$m ITERS=100 python tools/gemm_microbench.py 128 4096 3072 128 4096 1024 128 10000 1024 128 18192 1024 2560 4096 1024 2>&1 | grep -v amdgpu.ids
done
echo "== parity under SET_GEMM_SPLIT=2"
SET_GEMM_SPLIT=2 timeout 600 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_hip_editnet.py tests/test_hip_ops.py -k "full_b128 or full_b4 or v9490 or linear_shapes or token_table" 2>&1 | tail -3
echo "== bench A/B"
for rep in 1 2; do for m in 0 2 3 1; do
  SET_GEMM_SPLIT=$m python bench.py --no-cpu-baseline --no-train --no-secondary --repeat 2 --steps ${AB_STEPS:-100} --streams 7 > gpurun_out/r6/split_ab.json 2> gpurun_out/r6/split_ab.err
  python - $m <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6/split_ab.json"))
k = d["kernels"]
sel = {n: round(1e3 * k[n]["ms_per_step"] / max(k[n]["launches_per_step"], 1), 2) for n in k if "gemm_nt" in n}
print("[SET_GEMM_SPLIT=%s] value %.0f  single %.0f  us/launch %s" % (sys.argv[1], d["value"], d.get("single_stream_decode_steps_per_sec") or 0, sel))
PY
done; done
