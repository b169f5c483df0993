set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_hip_beam.py tests/test_hip_train.py tests/test_hip_train_mode.py tests/test_hip_sequence.py tests/test_hip_sampling.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r5/t7_tests.log
tail -5 gpurun_out/r5/t7_tests.log
timeout 300 python -c "
import torch, json, sys
sys.path.insert(0,'.')
from tools import secondary
print(json.dumps(secondary.beam(torch.device('cuda',0))))
" > gpurun_out/r5/t7_beam.json 2> gpurun_out/r5/t7_beam.err
tail -2 gpurun_out/r5/t7_beam.json
timeout 300 python tools/bench_train.py --steps 10 --warmup 3 2>&1 | grep ms_per | cut -c1-160 > gpurun_out/r5/t7_train.json
cat gpurun_out/r5/t7_train.json
