#!/bin/bash
# per-frame-pattern step at smaller launch sizes: does a step of <= 256 MB of tables (Infinity-Cache resident) pay for its ramps?
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
for F in 10000 5000 3334 2500 2000; do
  timeout -k 10 200 python bench.py --pattern-group 1 --frames $F --no-extra-legs --no-cpu-baseline --steps 30 --warmup 5 > $O/pf_$F.json 2> $O/pf_$F.err || exit 1
  python - <<PY
import json
d=json.load(open("$O/pf_$F.json")); r=d["roofline"]
print($F, "ms/step %.3f"%d["ms_per_step"], "per10k %.3f"%(d["ms_per_step"]*10000/$F), "pattern %.3f frame %.3f taps %.3f"%(d["config"].get("plan_build_kernel_ms",0), r["kernel_ms"], r["taps_kernel_ms"]), "step_frac %.3f"%r["step_frac"])
PY
done
