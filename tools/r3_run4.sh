#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests/test_gpu_patterns.py tests/test_gpu_edges.py tests/test_gpu_dataset.py tests/test_gpu_statistics.py -x -q > $O/t_patterns.log 2>&1; echo "tests rc=$?"; tail -5 $O/t_patterns.log
timeout -k 10 200 python tools/bench_patterns.py > $O/bench_patterns.txt 2>&1; echo "bench_patterns rc=$?"; cat $O/bench_patterns.txt
timeout -k 10 200 python tools/stamps_patterns.py > $O/stamps_pattern_kernel.txt 2>&1; echo "stamps rc=$?"; grep -E "total|selection|slot|row|strips|pockets|flips|plan" $O/stamps_pattern_kernel.txt
bash tools/pmc_pattern_phases.sh 838 > $O/pmc_pattern_phases_P838.txt 2>&1; grep -A7 "SQ_INSTS_VALU\|GRBM" $O/pmc_pattern_phases_P838.txt
