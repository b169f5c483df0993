#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
for q in 4 8 16; do
 for s in 7 11 16; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 150 --streams $s --no-cpu-baseline --no-secondary --no-train --no-profile > gpurun_out/p17.json 2>/dev/null
  python - $q $s <<'PY'
import json, sys
d=json.loads(open("gpurun_out/p17.json").read().strip().splitlines()[-1])
print("queues", sys.argv[1], "streams", sys.argv[2], d["value"], d.get("single_stream_decode_steps_per_sec"))
PY
 done
done
