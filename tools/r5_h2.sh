#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2000 python -m pytest tests/test_hip_adaptive.py tests/test_hip_sequence.py tests/test_hip_train.py -x -q -m gpu 2>&1 | tail -3
python - <<'PY'
import torch, json, sys, os
sys.path.insert(0, ".")
from tools import secondary
print(json.dumps(secondary.adaptive(torch.device("cuda:0"))))
PY
SET_XE_C_LOOPS=0 SET_SLAB_DIRECT=0 python - <<'PY'
import torch, json, sys, os
sys.path.insert(0, ".")
from tools import secondary
print("old path", json.dumps(secondary.adaptive(torch.device("cuda:0"))))
PY
