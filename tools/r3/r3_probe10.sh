#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
python tools/bench_dcnet.py 2>&1 | grep -v amdgpu.ids | tail -8
SET_ENC_PERSISTENT=0 python tools/bench_dcnet.py 2>&1 | grep -v amdgpu.ids | tail -8
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-train > gpurun_out/p10.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/p10.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","steps","single_stream_decode_steps_per_sec","single_stream_ms_per_step","batches_in_flight_per_gpu")}, d["repeat"], d["config"]["xe_forward_single_stream_decode_steps_per_sec"])
PY
