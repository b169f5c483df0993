set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_hip_persistent_decode.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r5/t4_tests.log
tail -5 gpurun_out/r5/t4_tests.log
timeout 300 python -c "
import torch, json, sys
sys.path.insert(0,'.')
from tools import secondary
print(json.dumps(secondary.batch_sweep(torch.device('cuda',0), batches=(4,5,6,8,12,16,32))))
" > gpurun_out/r5/t4_sweep.json 2> gpurun_out/r5/t4_sweep.err
SET_PDEC_STAMPS=2 timeout 120 python tools/profile_small_batch.py 16 2>&1 | tail -14 > gpurun_out/r5/t4_stamps16.txt
SET_PDEC_STAMPS=2 timeout 120 python tools/profile_small_batch.py 8 2>&1 | tail -14 > gpurun_out/r5/t4_stamps8.txt
timeout 300 python tools/bench_train.py --steps 10 --warmup 3 > gpurun_out/r5/t4_train.json 2>&1
timeout 300 python tools/bench_wgrad_shapes.py > gpurun_out/r5/t4_wgrad.txt 2>&1
