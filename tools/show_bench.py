#!/usr/bin/env python3
"""One-screen summary of a bench.py JSON line (file argument)."""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print(f"main  {d['config']['workload']}: {d['value']/1e6:.3f} M frames/s  {d['ms_per_step']:.3f} ms/step  kernel {r['kernel_ms']:.3f} ms frac {r['frac']:.3f} step_frac {r['step_frac']:.3f} traffic {r.get('traffic')} tri {d['config'].get('triangulation')}")
p = d.get("per_frame_patterns")
if p and "value" in p:
    print(f"per-frame: {p['value']/1e6:.3f} M  {p['ms_per_step']:.3f} ms  frac {p['frac']:.3f}  pattern {p['pattern_kernel_ms']:.3f} frame {p['frame_kernel_ms']:.3f} taps {p['taps_kernel_ms']:.3f}")
for k, v in (d.get("also") or {}).items():
    if "value" in v and "step_frac" in v:
        print(f"  {k}: {v['value']/1e6:.3f} M  {v['ms_per_step']:.3f} ms  frac {v['frac']:.3f} step_frac {v['step_frac']:.3f} kernel_ms {v.get('kernel_ms'):.3f} {v.get('kernels_ms','')} {v.get('mfma','')}")
    elif "value" in v:
        print(f"  {k}: {v['value']/1e3:.0f} k/s  {v['ms_per_step']:.3f} ms/step  {v.get('bound')} frac {v.get('frac', 0):.3f}  ({v.get('unit')})")
    else:
        print(f"  {k}: " + ", ".join(f"{a} {b:.3f}" for a, b in v.items() if isinstance(b, float)))
if "cpu_baseline" in d:
    c = d["cpu_baseline"]; print(f"cpu: {c['value']:.1f} {c['unit']} on {c['cores']} cores ({c['kind']}) ref_ratio {c.get('ref_ratio')}")
