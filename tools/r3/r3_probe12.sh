#!/bin/bash
# pick + next attention-LSTM cell in one launch (LstmTail): parity, then A/B against the separate pointwise launch
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for v in 0 1 0 1; do
  SET_PICK_TAIL=$v python bench.py --steps 150 --no-cpu-baseline --no-secondary --no-train > gpurun_out/p12_$v.json 2>/dev/null
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
d=json.loads(open("gpurun_out/p12_%s.json" % v).read().strip().splitlines()[-1])
print("[SET_PICK_TAIL=%s]" % v, {k:d.get(k) for k in ("value","single_stream_decode_steps_per_sec","batches_in_flight_per_gpu")},
      {k:v_["us_per_launch"] for k,v_ in d["kernels"].items() if k in ("greedy_pick","lstm_pointwise")},
      {k:v_["launches_per_step"] for k,v_ in d["kernels"].items() if k in ("greedy_pick","lstm_pointwise")})
PY
done
for v in 0 1; do
  echo "== B=4 SET_PICK_TAIL=$v"
  SET_PICK_TAIL=$v python tools/profile_small_batch.py 4 2>&1 | grep -v amdgpu.ids | head -8
  SET_PICK_TAIL=$v python tools/bench_dcnet.py 2>&1 | grep -v amdgpu.ids | tail -4
done
