#!/bin/bash
# the driver's command (--steps 20): how the 20 decodes of a window spread over the batches in flight
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
for s in 4 5 7 10 11 20 5 10 20; do
  python bench.py --steps 20 --warmup 10 --streams $s --no-cpu-baseline --no-secondary --no-train --no-profile > gpurun_out/p20.json 2>/dev/null
  python - $s <<'PY'
import json, sys
d=json.loads(open("gpurun_out/p20.json").read().strip().splitlines()[-1])
print("streams", sys.argv[1], d["value"], d["repeat"]["min"], d["repeat"]["max"])
PY
done
