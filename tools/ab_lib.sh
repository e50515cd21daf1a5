#!/bin/bash
# A/B of two builds of the library on the bench workload, alternating processes: bash tools/ab_lib.sh <other.so> [bench args]
other=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
show() { python3 -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); ro=r['roofline']
print('$1', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'kernel %.3f ms' % (ro.get('kernel_ms') or 0), 'frac %.3f' % ro['frac'], 'step_frac %.3f' % (ro.get('step_frac') or 0), 'pattern %.3f ms' % (r['config'].get('plan_build_kernel_ms') or 0), ro.get('kernels_ms') or '')"; }
for i in 1 2 3; do
  timeout -k 10 200 python $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | show base || exit 1
  CHEST_LIB=$other timeout -k 10 200 python $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | show other || exit 1
done
