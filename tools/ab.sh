#!/bin/bash
# A/B of compile-time variants on ONE GPU box: tools/ab.sh "<flags A>" "<flags B>" ...   (each arg = SET_HIPCC_FLAGS)
# prints headline / single-stream / selected kernel timings of bench.py for every variant, twice (ABAB order).
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for flags in "$@"; do
  SET_HIPCC_FLAGS="$flags" python -m show_edit_tell_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  python bench.py --no-cpu-baseline --no-train --no-secondary --repeat 2 --steps ${AB_STEPS:-150} --streams ${AB_STREAMS:-7} > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - "$flags" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
k = d["kernels"]
sel = {n: round(1e3 * k[n]["ms_per_step"] / max(k[n]["launches_per_step"], 1), 2) for n in k if any(x in n for x in ("attention", "copy_gate", "encoder", "pointwise", "pick", "gemm_nt_f32<64"))}
print("[%s] value %.0f  single %.0f  us/launch %s" % (sys.argv[1], d["value"], (d.get("single_stream_decode_steps_per_sec") or d["value"]), sel))
PY
done
done
python -m show_edit_tell_amd.build --force > /dev/null 2>&1
