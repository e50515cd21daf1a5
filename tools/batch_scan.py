#!/usr/bin/env python3
"""Frame-kernel time against batch size on ONE placed output set (smaller batches write into the leading part of the same
tensors): separates a per-launch cost from the per-byte cost."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
E = importlib.import_module("channel-estimation-in-5g-network_amd.engine")
geom = E.Geometry()
eng = E.Engine(geom, 0)
G, F = geom.grid, 20000
slots = [eng.pattern_slot(np.sort(np.random.RandomState(g).permutation(G)[: G // 10]), "linear") for g in range(20)]
d = E.make_descs("EVA", 111.1, 10.0, np.asarray(slots, np.int32)[np.arange(F) // 1024], n=F)
names = ["tx", "rx", "H", "H_ls", "H_mmse", "pilot_symbols", "pilot_mask"]
out, rep = eng.place_outputs(d, names, seed=1, tries=4)
print("placement (20 000 frames):", rep)
sizes = [625, 1250, 2500, 5000, 10000, 20000]
dev = {n: torch.from_numpy(d[:n].view(np.uint8).copy()).to(eng.device) for n in sizes}
t = {n: [] for n in sizes}
tt = {n: [] for n in sizes}
for it in range(8):
    for n in sizes:
        o = {k: v[:n] for k, v in out.items()}
        eng.profile_begin(1)
        eng.run_frames(dev[n], seed=1, frame_offset=it * F, outputs=names, out=o, check=False)
        a, b, _ = eng.profile_read()
        if it >= 2:
            t[n].append(b); tt[n].append(a)
xs = np.array(sizes, float); ys = np.array([np.mean(t[n]) for n in sizes])
slope, icpt = np.polyfit(xs[2:], ys[2:], 1)
for n in sizes:
    print(f"{n:6d} frames: frame_kernel {np.mean(t[n]):7.3f} ms  taps_kernel {np.mean(tt[n]):6.3f} ms   {n * 1088498 / np.mean(t[n]) * 1e-9:6.2f} TB/s")
print(f"fit over >= 2 500 frames: {icpt:.3f} ms + {slope * 1e3:.4f} us/frame  (= {1088498 / slope * 1e-9:.2f} TB/s marginal)")
