#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m show_edit_tell_amd.build > /dev/null 2>&1
sum() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], {k:d.get(k) for k in ("value","steps","single_stream_decode_steps_per_sec","single_stream_ms_per_step","batches_in_flight_per_gpu")}, d["repeat"])
PY
}
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-train > gpurun_out/p9_head_$i.json 2>/dev/null; sum gpurun_out/p9_head_$i.json; done
SET_BENCH_MIN_WINDOWS=5 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-train > gpurun_out/p9_head_w5.json 2>/dev/null; sum gpurun_out/p9_head_w5.json
python bench.py --gpus 1 --steps 20 --warmup 5 --streams 7 --no-cpu-baseline --no-secondary --no-train > gpurun_out/p9_head_s7.json 2>/dev/null; sum gpurun_out/p9_head_s7.json
( cd .old_r02 && python -m show_edit_tell_amd.build > /dev/null 2>&1; for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-train > ../gpurun_out/p9_old_$i.json 2>/dev/null; done )
sum gpurun_out/p9_old_1.json; sum gpurun_out/p9_old_2.json
