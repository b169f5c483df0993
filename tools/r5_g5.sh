set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r5/t5_tests.log
tail -5 gpurun_out/r5/t5_tests.log
timeout 300 python -c "
import torch, json, sys
sys.path.insert(0,'.')
from tools import secondary
print(json.dumps(secondary.batch_sweep(torch.device('cuda',0))))
" > gpurun_out/r5/t5_sweep.json 2> gpurun_out/r5/t5_sweep.err
timeout 300 python tools/bench_train.py --steps 10 --warmup 3 > gpurun_out/r5/t5_train.json 2>&1
timeout 300 python tools/bench_beam.py > gpurun_out/r5/t5_beam.json 2>&1
