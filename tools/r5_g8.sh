set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_hip_beam.py tests/test_hip_sequence.py -m gpu -q 2>&1 | tail -5 > gpurun_out/r5/t8_tests.log
tail -3 gpurun_out/r5/t8_tests.log
SET_PDEC_STAMPS=2 timeout 300 python -c "
import torch, json, sys
sys.path.insert(0,'.')
from tools import secondary
print(json.dumps(secondary.beam(torch.device('cuda',0))))
" > gpurun_out/r5/t8_beam.json 2> gpurun_out/r5/t8_beam.err
tail -1 gpurun_out/r5/t8_beam.json
grep "pdec stamps" gpurun_out/r5/t8_beam.err | tail -3
