#!/bin/bash
# Which write level is this box on, and what do its clocks say?  (read-only rocm-smi queries around a short bench run)
#   bash tools/box_level.sh        (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
rocm-smi -s 2>/dev/null | grep -E "fclk|mclk|Supported|^GPU.*: [0-9]+:" | head -30
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "fclk|mclk|sclk|Power \(W\)" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/\t//g; s/  */ /g'; echo; sleep 0.1; done ) > /tmp/smi_during.txt &
timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-extra-legs --steps 400 --warmup 10 2>/dev/null | python3 -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); ro=r['roofline']
print('bench: %.3f M frames/s, frame kernel %.3f ms = %.3f of 8 TB/s' % (r['value']/1e6, ro['kernel_ms'], ro['frac']))"
wait
echo "rocm-smi while the bench ran (samples with more than 600 W):"; grep -E "Power \(W\): ([6-9][0-9][0-9]|1[0-9][0-9][0-9])" /tmp/smi_during.txt | sort | uniq -c | sort -rn | head -8
