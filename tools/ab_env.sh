#!/bin/bash
# A/B of RUNTIME knobs on ONE GPU box: tools/ab_env.sh "<ENV=.. ENV=..>" "<...>" ...  (ABAB order, two rounds)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg python bench.py --no-cpu-baseline --no-train --no-secondary --repeat 2 --steps ${AB_STEPS:-150} --streams ${AB_STREAMS:-7} > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
k = d["kernels"]
sel = {n: round(1e3 * k[n]["ms_per_step"] / max(k[n]["launches_per_step"], 1), 2) for n in k if any(x in n for x in ("attention", "copy_gate", "encoder", "pointwise", "pick", "gemm_nt_f32"))}
print("[%s] value %.0f  single %.0f  frac %s us/launch %s" % (sys.argv[1], d["value"], (d.get("single_stream_decode_steps_per_sec") or d["value"]), d.get("roofline", {}).get("frac"), sel))
PY
done
done
