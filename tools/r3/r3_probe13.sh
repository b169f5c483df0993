#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
timeout 900 python -m pytest tests/test_hip_editnet.py tests/test_hip_dcnet.py tests/test_hip_sampling.py tests/test_hip_atsize.py tests/test_hip_properties.py -m gpu -q -x 2>&1 | tail -3
for v in 0 1 0 1; do
  SET_PICK_TAIL=$v python bench.py --steps 150 --no-cpu-baseline --no-secondary --no-train > gpurun_out/p13_$v.json 2>/dev/null
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
d=json.loads(open("gpurun_out/p13_%s.json" % v).read().strip().splitlines()[-1])
print("[SET_PICK_TAIL=%s]" % v, {k:d.get(k) for k in ("value","single_stream_decode_steps_per_sec","batches_in_flight_per_gpu")},
      {k:v_["us_per_launch"] for k,v_ in d["kernels"].items() if k in ("greedy_pick","lstm_pointwise")})
PY
done
for v in 0 1; do
  echo "== SET_PICK_TAIL=$v"
  SET_PICK_TAIL=$v python tools/bench_dcnet.py 2>&1 | grep -v amdgpu.ids | tail -2
done
