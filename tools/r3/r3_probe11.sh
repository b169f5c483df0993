#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python bench.py --steps 150 --no-cpu-baseline --no-secondary --no-train > gpurun_out/p11.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/p11.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","steps","single_stream_decode_steps_per_sec","single_stream_ms_per_step","batches_in_flight_per_gpu")}, d["repeat"])
print({k:v["us_per_launch"] for k,v in d["kernels"].items()})
print(d["roofline"].get("traffic_step_launches"))
PY
python tools/profile_small_batch.py 4 2>&1 | grep -v amdgpu.ids | head -3
