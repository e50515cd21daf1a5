#!/bin/bash
# full GPU suite + the driver's bench invocation
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -15 $O/t_gpu.log
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 6000 $O/bench_default.json; tail -5 $O/bench_default.err
