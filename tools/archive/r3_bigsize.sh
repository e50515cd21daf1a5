#!/bin/bash
# config 4 at one rank's real shard size (125 000 frames resident = 138 GB) and the footprint sweep of the frame kernel
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --workload cfg4_dataset_mixed --pattern-group 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg4_pf_10000.json 2> $O/bench_cfg4_pf_10000.err; echo "cfg4 10000 rc=$?"
timeout -k 10 500 python bench.py --workload cfg4_dataset_mixed --pattern-group 1 --frames 125000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg4_pf_125000.json 2> $O/bench_cfg4_pf_125000.err; echo "cfg4 125000 rc=$?"; tail -3 $O/bench_cfg4_pf_125000.err
timeout -k 10 500 python bench.py --workload cfg4_dataset_mixed --frames 125000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg4_shared_125000.json 2> $O/bench_cfg4_shared_125000.err; echo "cfg4 shared 125000 rc=$?"
timeout -k 10 600 python tools/footprint_sweep.py --no-pcie > $O/footprint_sweep.txt 2>&1; echo "sweep rc=$?"; cat $O/footprint_sweep.txt
rocprofv3 -L 2>/dev/null | grep -i -E "utcl2|tlb|translat" | head -40 > $O/tlb_counters.txt; cat $O/tlb_counters.txt | head -30
for f in $O/bench_cfg4_pf_10000.json $O/bench_cfg4_pf_125000.json $O/bench_cfg4_shared_125000.json; do python - "$f" <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); c=r["config"]; ro=r["roofline"]
    print(sys.argv[1].split('/')[-1], "frames", c["frames_per_step_per_gpu"], "value %.3f M/s"%(r["value"]/1e6), "ms/step %.3f"%r["ms_per_step"], "frame kernel %.3f ms frac %.3f step_frac %.3f"%(ro["kernel_ms"], ro["frac"], ro["step_frac"]), "pattern kernel", c.get("plan_build_kernel_ms"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done
