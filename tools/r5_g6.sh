set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_hip_beam.py tests/test_hip_persistent_decode.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r5/t6_tests.log
tail -5 gpurun_out/r5/t6_tests.log
timeout 300 python -c "
import torch, json, sys
sys.path.insert(0,'.')
from tools import secondary
print(json.dumps(secondary.beam(torch.device('cuda',0))))
" > gpurun_out/r5/t6_beam.json 2> gpurun_out/r5/t6_beam.err
tail -3 gpurun_out/r5/t6_beam.json
