#!/bin/bash
# round 3, first GPU call: the rebuilt flip stage of the pattern kernel (tests, rate, stamps, per-frame step)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_patterns.py -x -q > $O/t_patterns.log 2>&1; echo "patterns tests rc=$?"; tail -3 $O/t_patterns.log
timeout -k 10 200 python tools/bench_patterns.py > $O/bench_patterns.txt 2>&1; echo "bench_patterns rc=$?"; cat $O/bench_patterns.txt
timeout -k 10 200 python tools/stamps_patterns.py > $O/stamps_pattern_kernel.txt 2>&1; echo "stamps rc=$?"; cat $O/stamps_pattern_kernel.txt
timeout -k 10 300 python bench.py --pattern-group 1 --no-cpu-baseline > $O/bench_cfg2_pf.json 2> $O/bench_cfg2_pf.err; echo "bench pf rc=$?"; cat $O/bench_cfg2_pf.json
