#!/usr/bin/env bash
# Multi-GPU check of the one exchange step of the path (the gradient all-reduce of XE / SCST training, hook between
# /root/reference editnet.py:579 and :580) — ONE command on a node with more than one MI355X:
#
#     bash tools/scale_check.sh [ROUND]          ->  profiles/rNN_scale.json  (+ gpurun_out/scale/*.log)
#
# 1. the two `nccl` tests that a 1-GPU box skips (tests/test_hip_dp.py: two ranks on two devices, bench.py --gpus 2);
# 2. bench.py --gpus {1,2,4,8} exactly as the driver launches it (torch.distributed.run, 127.0.0.1), keeping from every
#    line: decode-steps/s, and of the training leg ms per step with / without the collective, the collective's own time,
#    exposed vs overlapped time, the bucket sizes, which device / backend every rank ran on (`ranks_seen`), RCCL's version
#    and the xGMI topology (`fabric`).
# On a 1-GPU box it says so and writes a record with "skipped".  Nothing here reads /root/reference.
set -u
cd "$(dirname "$0")/.."
ROUND=${1:-05}
OUT=profiles/r${ROUND}_scale.json
LOG=gpurun_out/scale
mkdir -p "$LOG" profiles
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
export NCCL_DEBUG=${NCCL_DEBUG:-VERSION}
NDEV=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "devices: $NDEV"
if [ "$NDEV" -lt 2 ]; then
    python - "$OUT" "$NDEV" <<'PY'
import json, sys
json.dump({"skipped": True, "reason": "torch.cuda.device_count() = %s: the exchange step needs two devices" % sys.argv[2]},
          open(sys.argv[1], "w"), indent=1)
print(open(sys.argv[1]).read())
PY
    exit 0
fi
timeout 1800 python -m pytest tests/test_hip_dp.py -m gpu -q -k "nccl" > "$LOG/nccl_tests.log" 2>&1
echo "nccl tests rc=$?" | tee -a "$LOG/nccl_tests.log"
PORT=29770
for N in 1 2 4 8; do
    [ "$N" -gt "$NDEV" ] && continue
    if [ "$N" -eq 1 ]; then
        timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > "$LOG/bench_n$N.json" 2> "$LOG/bench_n$N.err"
    else
        timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((PORT + N)) \
            bench.py --gpus "$N" --steps 20 --warmup 5 > "$LOG/bench_n$N.json" 2> "$LOG/bench_n$N.err"
    fi
    echo "bench --gpus $N rc=$?"
done
python - "$OUT" "$LOG" <<'PY'
import json, os, sys
out, log = sys.argv[1], sys.argv[2]
rec = {"skipped": False, "runs": []}
try:
    rec["nccl_tests_tail"] = open(os.path.join(log, "nccl_tests.log")).read().strip().splitlines()[-3:]
except OSError:
    pass
base = None
for n in (1, 2, 4, 8):
    f = os.path.join(log, "bench_n%d.json" % n)
    if not os.path.exists(f):
        continue
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        rec["runs"].append({"n_gpus": n, "error": repr(e)[:200]})
        continue
    tr = d.get("train", {})
    if n == 1:
        base = d["value"]
    rec["runs"].append({"n_gpus": n, "decode_steps_per_sec": d["value"], "vs_1gpu": None if not base else round(d["value"] / base, 3),
                        "train_ms_per_step": tr.get("ms_per_train_step"), "train_ms_per_step_no_allreduce": tr.get("ms_per_train_step_no_allreduce"),
                        "allreduce_ms": tr.get("allreduce_ms"), "allreduce_exposed_ms": tr.get("allreduce_exposed_ms"),
                        "allreduce_overlapped_ms": None if tr.get("allreduce_ms") is None or tr.get("allreduce_exposed_ms") is None
                        else round(tr["allreduce_ms"] - tr["allreduce_exposed_ms"], 3),
                        "allreduce_buckets_MB": tr.get("allreduce_buckets"), "ranks_seen": tr.get("ranks_seen"), "fabric": tr.get("fabric")})
json.dump(rec, open(out, "w"), indent=1)
print(open(out).read()[:3000])
PY
