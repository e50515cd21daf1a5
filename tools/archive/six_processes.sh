for i in 1 2 3 4 5 6; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); ro=r['roofline']
print('process: %.3f M frames/s, step %.3f ms, frame kernel %.3f ms = %.3f, step_frac %.3f' % (r['value']/1e6, r['ms_per_step'], ro['kernel_ms'], ro['frac'], ro['step_frac']))"; done
