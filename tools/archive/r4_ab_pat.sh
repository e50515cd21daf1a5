#!/bin/bash
# pattern kernel A/B: base library (libchest_base.so at the repo root) against the working one, alternating processes; then the pattern tests
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in 1 2; do
  echo "--- base";  CHEST_LIB=$R/libchest_base.so timeout -k 10 200 python tools/bench_patterns.py 10000 2>/dev/null | grep -E "0.10|0.05" | head -7 || exit 1
  echo "--- new";   timeout -k 10 200 python tools/bench_patterns.py 10000 2>/dev/null | grep -E "0.10|0.05" | head -7 || exit 1
done
timeout -k 10 600 python -m pytest tests/test_gpu_patterns.py tests/test_gpu_dataset.py -x -q 2>&1 | tail -4
