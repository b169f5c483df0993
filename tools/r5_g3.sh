set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_hip_persistent_decode.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r5/t3_tests.log
tail -5 gpurun_out/r5/t3_tests.log
for nc in 0 1 2; do
SET_DEC_WIDE_COURIERS=$nc timeout 300 python -c "
import torch, json, sys
sys.path.insert(0,'.')
from tools import secondary
print(json.dumps(secondary.batch_sweep(torch.device('cuda',0), batches=(4,5,6,8,12,16))))
" > gpurun_out/r5/t3_sweep_nc$nc.json 2> gpurun_out/r5/t3_sweep_nc$nc.err
done
SET_DEC_WIDE_MINB=99 timeout 300 python -c "
import torch, json, sys
sys.path.insert(0,'.')
from tools import secondary
print(json.dumps(secondary.batch_sweep(torch.device('cuda',0), batches=(4,5,6,8,12,16))))
" > gpurun_out/r5/t3_sweep_old.json 2> gpurun_out/r5/t3_sweep_old.err
SET_PDEC_STAMPS=2 SET_DEC_WIDE_COURIERS=2 timeout 120 python tools/profile_small_batch.py 16 > gpurun_out/r5/t3_stamps_nc2.txt 2>&1
SET_PDEC_STAMPS=2 SET_DEC_WIDE_COURIERS=0 timeout 120 python tools/profile_small_batch.py 16 > gpurun_out/r5/t3_stamps_nc0.txt 2>&1
