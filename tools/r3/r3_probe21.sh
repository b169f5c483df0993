#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
for v in 1 2 1 2; do
  SET_GEMM_KGROUPS=$v python bench.py --steps 150 --no-cpu-baseline --no-secondary --no-train --no-profile > gpurun_out/p21.json 2>/dev/null
  python - $v <<'PY'
import json, sys
d=json.loads(open("gpurun_out/p21.json").read().strip().splitlines()[-1])
print("KGROUPS", sys.argv[1], d["value"], d["single_stream_decode_steps_per_sec"], d["batches_in_flight_per_gpu"], d["stream_probe_decode_steps_per_sec"])
PY
done
