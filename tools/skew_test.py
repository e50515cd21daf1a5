#!/usr/bin/env python3
"""Does the relative placement of the five output arrays matter (DRAM channel aliasing)?  All outputs are carved
from ONE buffer; array i starts at its natural offset + i*skew bytes (rounded to 256 B).  GPU only."""
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
G, F = geom.grid, 10000
slots = [eng.pattern_slot(np.sort(np.random.RandomState(g).permutation(G)[: G // 10]), "linear") for g in range(10)]
d = E.make_descs("EVA", 111.1, 10.0, np.asarray(slots, np.int32)[np.arange(F) // 1024], n=F)
d_dev = torch.from_numpy(d.view(np.uint8).copy()).to(eng.device)
shapes = {"tx": (F, 14, 2, 599), "rx": (F, 14, 2, 599), "H": (F, 14, 2, 2, 599), "H_ls": (F, 14, 2, 2, 599), "H_mmse": (F, 14, 2, 2, 599)}
sizes = {k: int(np.prod(v)) * 8 for k, v in shapes.items()}
total = sum(sizes.values())
buf = torch.empty(total + (64 << 20), dtype=torch.uint8, device=eng.device)
base = buf.data_ptr()
pad0 = (-base) % 4096


def views(skew, order=("tx", "rx", "H", "H_ls", "H_mmse")):
    out, off = {}, pad0
    for i, k in enumerate(order):
        o = (off + i * skew + 255) // 256 * 256
        n = sizes[k]
        out[k] = buf[o:o + n].view(torch.complex64).view(shapes[k])
        off = o + n
    return out


def timeit(out, names, iters=12):
    for _ in range(3):
        eng.run_frames(d_dev, outputs=names, out=out, check=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.run_frames(d_dev, outputs=names, out=out, check=False); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return np.median(ts), np.min(ts)


def views_aligned(align, extra=0):
    out, off = {}, (-base) % align
    for k in ("tx", "rx", "H", "H_ls", "H_mmse"):
        o = (off + align - 1) // align * align + extra
        out[k] = buf[o:o + sizes[k]].view(torch.complex64).view(shapes[k])
        off = o + sizes[k]
    return out


names = ("tx", "rx", "H", "H_ls", "H_mmse")
for align in (256, 4096, 65536, 1 << 20, 2 << 20):
    med, mn = timeit(views_aligned(align), names)
    print(f"one buffer, every array start aligned to {align:8d} B: median {med:6.3f} ms  min {mn:6.3f}", flush=True)
for extra in (16, 64, 128, 256, 1024):
    med, mn = timeit(views_aligned(2 << 20, extra), names)
    print(f"one buffer, 2 MiB-aligned + {extra:5d} B:                 median {med:6.3f} ms  min {mn:6.3f}", flush=True)
outs = {k: torch.empty(sizes[k] + 4096, dtype=torch.uint8, device=eng.device) for k in names}
for extra in (0, 16, 256, 2048):
    o = {k: outs[k][extra:extra + sizes[k]].view(torch.complex64).view(shapes[k]) for k in names}
    med, mn = timeit(o, names)
    print(f"separate allocations, start + {extra:5d} B:                median {med:6.3f} ms  min {mn:6.3f}", flush=True)
del outs
print("natural sizes mod 4096:", {k: v % 4096 for k, v in sizes.items()})
for skew in (0, 4096 + 256):
    med, mn = timeit(views(skew), names)
    print(f"skew {skew:8d} B   median {med:6.3f} ms   min {mn:6.3f} ms   {F / med / 1e3:6.3f} M frames/s", flush=True)
# separately allocated tensors (what the engine does by default), three fresh allocations
for rep in range(3):
    out = {k: torch.empty(shapes[k], dtype=torch.complex64, device=eng.device) for k in names}
    med, mn = timeit(out, names)
    print(f"separate torch allocations #{rep}: median {med:6.3f} ms   min {mn:6.3f} ms   ptr%65536 = {[out[k].data_ptr() % 65536 for k in names]}", flush=True)
    del out
