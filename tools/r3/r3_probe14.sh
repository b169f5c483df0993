#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
for v in 256 64 128 256 64 128; do
  SET_PW_BLOCK=$v python bench.py --steps 150 --no-cpu-baseline --no-secondary --no-train > gpurun_out/p14_$v.json 2>/dev/null
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
d=json.loads(open("gpurun_out/p14_%s.json" % v).read().strip().splitlines()[-1])
print("[SET_PW_BLOCK=%s]" % v, {k:d.get(k) for k in ("value","single_stream_decode_steps_per_sec","batches_in_flight_per_gpu")},
      {k:v_["us_per_launch"] for k,v_ in d["kernels"].items() if k in ("greedy_pick","lstm_pointwise")})
PY
done
