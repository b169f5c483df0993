set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5/t1_tests.log
timeout 300 python -c "
import torch, json, sys
sys.path.insert(0,'.')
from tools import secondary
print(json.dumps(secondary.batch_sweep(torch.device('cuda',0)), indent=1))
" > gpurun_out/r5/t1_sweep.json 2> gpurun_out/r5/t1_sweep.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5/t1_bench.json 2> gpurun_out/r5/t1_bench.err
tail -3 gpurun_out/r5/t1_tests.log
