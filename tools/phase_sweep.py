#!/usr/bin/env python3
"""Does the RELATIVE address phase of the five output tensors matter?  A workgroup writes its frame's slices of tx, rx,
H, H_ls and H_mmse at the same time; if the tensors start at equal offsets modulo the DRAM interleave, those concurrent
streams keep meeting in the same channel / bank.  Each tensor gets its own 4 MiB-aligned region inside one buffer and
starts k * delta bytes into it (k = 0..4).  In-process, interleaved, kernel time from the library's HIP events."""
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
names = ("tx", "rx", "H", "H_ls", "H_mmse")
shapes = {"tx": (F, 14, 2, 599), "rx": (F, 14, 2, 599), "H": (F, 14, 2, 2, 599), "H_ls": (F, 14, 2, 2, 599), "H_mmse": (F, 14, 2, 2, 599)}
sizes = {k: int(np.prod(v)) * 8 for k, v in shapes.items()}
REGION = 4 << 20
MAXD = 80 << 20
buf = torch.empty(sum(sizes.values()) + 5 * (REGION + 5 * MAXD), dtype=torch.uint8, device=eng.device)
base = buf.data_ptr()


def views(delta):
    out, off = {}, (-base) % REGION
    for i, k in enumerate(names):
        o = off + i * delta
        assert o % 16 == 0
        out[k] = buf[o:o + sizes[k]].view(torch.complex64).view(shapes[k])
        off = (o + sizes[k] + REGION - 1) // REGION * REGION
    return out


if '--quick' in sys.argv:
    deltas = [0, (33 << 20) + 65536 + 256]
else:
  deltas = [0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 832 * 1024 + 256, 3 * 4096 + 512, 7 * 65536 + 4096 + 256,
            17 << 20, (33 << 20) + 65536 + 256]
vs = {dl: views(dl) for dl in deltas}
t = {dl: [] for dl in deltas}
for i in range(10):
    for dl in deltas:
        eng.profile_begin(1)
        eng.run_frames(d_dev, seed=1, frame_offset=i * F, outputs=names, out=vs[dl], check=False)
        _, k_ms, _ = eng.profile_read()
        if i >= 2:
            t[dl].append(k_ms)
for dl in deltas:
    a = np.array(t[dl])
    print(f"tensor k starts k * {dl:9d} B into its 4 MiB-aligned region: frame_kernel {a.mean():6.3f} ms (sd {a.std():.3f})  {10.73 / a.mean():5.2f} TB/s", flush=True)
# torch's own allocations, for reference
out = {k: torch.empty(shapes[k], dtype=torch.complex64, device=eng.device) for k in names}
print("separate torch allocations: base addresses mod 4 MiB:", {k: out[k].data_ptr() % REGION for k in names})
ts = []
for i in range(10):
    eng.profile_begin(1)
    eng.run_frames(d_dev, seed=1, frame_offset=i * F, outputs=names, out=out, check=False)
    ts.append(eng.profile_read()[1])
print(f"separate torch allocations: frame_kernel {np.mean(ts[2:]):6.3f} ms")
