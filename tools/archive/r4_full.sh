#!/bin/bash
# full GPU suite + the driver's bench invocation (round 4)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -6 $O/t_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; echo "bench rc=$rc"; wc -c $O/bench_default.json; tail -3 $O/bench_default.err
exit $rc
