#!/usr/bin/env python3
"""Which pairs of (synthesis tensors, estimate tensors) allocations sit at the slow write level?  K candidates per group, every
combination timed (Engine.place_outputs' own probe), with the byte addresses of the five large tensors -- raw material for a
placement rule.   python tools/placement_combos.py [K=5]"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
E = importlib.import_module("channel-estimation-in-5g-network_amd.engine")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
F = 10000
eng = E.Engine(E.Geometry(), 0)
G = eng.geom.grid
slots = eng.build_patterns(0, np.full(10, int(G * 0.1), np.int32), seed=2024, pattern_ids=np.arange(10, dtype=np.uint64), method="linear")
descs = E.make_descs("EVA", 111.1, 10.0, np.asarray(slots, np.int32)[np.arange(F) // 1000], n=F)
d_dev = torch.from_numpy(descs.view(np.uint8).copy()).to(eng.device)
syn, est = ["tx", "rx", "H", "pilot_symbols", "pilot_mask"], ["H_ls", "H_mmse"]
A = [eng.run_frames(d_dev, seed=1, outputs=syn) for _ in range(K)]
B = [eng.run_frames(d_dev, seed=1, outputs=est) for _ in range(K)]
names = syn + est
for _ in range(30):
    eng.run_frames(d_dev, seed=1, outputs=names, out={**A[0], **B[0]}, check=False)
rec = []
for rnd in range(2):
    for i in range(K):
        for j in range(K):
            out = {**A[i], **B[j]}
            eng.run_frames(d_dev, seed=1, outputs=names, out=out, check=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                eng.run_frames(d_dev, seed=1, outputs=names, out=out, check=False)
            e1.record(); e1.synchronize()
            rec.append((rnd, i, j, e0.elapsed_time(e1) / 5))
print(json.dumps({"A": [{k: a[k].data_ptr() for k in ("tx", "rx", "H")} for a in A], "B": [{k: b[k].data_ptr() for k in est} for b in B],
                  "ms": rec}))
tab = np.zeros((2, K, K))
for rnd, i, j, ms in rec:
    tab[rnd, i, j] = ms
print("step ms, rows = synthesis set i, columns = estimate set j (mean of two passes):")
for i in range(K):
    print("  " + "  ".join(f"{v:.3f}" for v in tab.mean(0)[i]))
