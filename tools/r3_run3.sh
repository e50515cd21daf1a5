#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_patterns.py tests/test_gpu_edges.py -x -q > $O/t_patterns.log 2>&1; echo "patterns+edges tests rc=$?"; tail -3 $O/t_patterns.log
timeout -k 10 200 python tools/bench_patterns.py > $O/bench_patterns.txt 2>&1; echo "bench_patterns rc=$?"; cat $O/bench_patterns.txt
timeout -k 10 200 python tools/stamps_patterns.py > $O/stamps_pattern_kernel.txt 2>&1; echo "stamps rc=$?"; cat $O/stamps_pattern_kernel.txt
timeout -k 10 300 python bench.py --pattern-group 1 --no-cpu-baseline --steps 10 > $O/bench_cfg2_pf.json 2> $O/bench_cfg2_pf.err; echo "bench pf rc=$?"
timeout -k 10 400 python bench.py --workload cfg5_4x4_2048_fullcov --frames 12500 --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_cfg5_12500.json 2> $O/bench_cfg5_12500.err; echo "cfg5 12500 rc=$?"
for f in $O/bench_cfg2_pf.json $O/bench_cfg5_12500.json; do python - "$f" <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); c=r["config"]; ro=r["roofline"]
    print(sys.argv[1].split('/')[-1], "frames", c["frames_per_step_per_gpu"], "value %.3f M/s"%(r["value"]/1e6), "ms/step %.3f"%r["ms_per_step"], "kernel %.3f ms frac %.3f step_frac %.3f"%(ro["kernel_ms"], ro["frac"], ro["step_frac"]), "pattern kernel", c.get("plan_build_kernel_ms"), ro.get("kernels_ms"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
