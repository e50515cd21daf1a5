#!/bin/bash
# config 5: the step as a pipeline of chunks over three streams
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_edges.py -x -q -k "full_covariance" > $O/t_fullcov.log 2>&1; echo "fullcov tests rc=$?"; tail -2 $O/t_fullcov.log
for c in 1 2 4 8; do
  timeout -k 10 300 python bench.py --workload cfg5_4x4_2048_fullcov --pipeline-chunks $c --no-cpu-baseline --steps 10 > $O/bench_cfg5_chunks$c.json 2> $O/bench_cfg5_chunks$c.err; echo "chunks $c rc=$?"
  python - $O/bench_cfg5_chunks$c.json <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); ro=r["roofline"]
    print(" chunks", r["config"]["pipeline_chunks"], "value %.1f k/s"%(r["value"]/1e3), "ms/step %.3f"%r["ms_per_step"], "step_frac %.3f"%ro["step_frac"], "kernels", {k: round(v,3) for k,v in ro["kernels_ms"].items()}, "mse", r["est_mse"])
except Exception as e: print("unreadable", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
