#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2000 python -m pytest tests/test_hip_persistent_decode.py tests/test_hip_editnet.py tests/test_hip_beam.py tests/test_hip_boundary.py -x -q -m gpu 2>&1 | tail -3
for f in 1 0 1 0; do echo "fork=$f"; SET_PROLOGUE_FORK=$f python - <<'PY'
import torch, json, sys
sys.path.insert(0, ".")
from tools import secondary
r = secondary.batch_sweep(torch.device("cuda:0"), batches=(1, 4, 8, 16))
print([ (x["batch"], x["ms_per_decode"]) for x in r["rows"]])
PY
done
