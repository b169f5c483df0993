cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
for m in 1 5; do
SET_DEC_WIDE_MINB=$m timeout 300 python -c "
import torch, json, sys
sys.path.insert(0,'.')
from tools import secondary
d=secondary.batch_sweep(torch.device('cuda',0), batches=(1,2,3,4))
print('minb=$m', [(r['batch'], r['ms_per_decode'], r['path']) for r in d['rows']])
" 2>&1 | grep minb
done
for b in 4; do SET_DEC_WIDE_MINB=1 SET_PDEC_STAMPS=2 python tools/profile_small_batch.py $b 2>&1 | grep "pdec stamps" | tail -1; SET_PDEC_STAMPS=2 python tools/profile_small_batch.py $b 2>&1 | grep "pdec stamps" | tail -1; done
