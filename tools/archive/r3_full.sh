#!/bin/bash
# full GPU suite + the driver's bench invocation + a two-rank gloo rehearsal of bench.py on one GPU
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/t_gpu.log
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; wc -l $O/bench_default.json
timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --no-cpu-baseline --frames 4000 > $O/bench_cfg2_gloo2_one_gpu.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"; wc -l $O/bench_cfg2_gloo2_one_gpu.json; head -c 300 $O/bench_cfg2_gloo2_one_gpu.json
