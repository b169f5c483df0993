cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_hip_persistent_decode.py tests/test_hip_beam.py tests/test_hip_dcnet.py tests/test_hip_editnet.py tests/test_hip_boundary.py -m gpu -q 2>&1 | tail -12 > gpurun_out/r5/t10_tests.log
tail -4 gpurun_out/r5/t10_tests.log
timeout 300 python -c "
import torch, json, sys
sys.path.insert(0,'.')
from tools import secondary
d=secondary.batch_sweep(torch.device('cuda',0), batches=(1,2,4,8,16))
print('editnet', [(r['batch'], r['ms_per_decode'], r['path']) for r in d['rows']])
print(json.dumps(secondary.dcnet(torch.device('cuda',0))))
" 2>&1 | grep -v amdgpu | cut -c1-1500
