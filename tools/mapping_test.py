#!/usr/bin/env python3
"""Frame -> XCD mapping variants on several output placements in one process: is there a mapping whose time does not
depend on where the outputs live?  CHEST_XCD_BLOCK=n gives XCD x blocks of n consecutive frames round-robin (n=1 is the
identity mapping; unset = one contiguous eighth of the batch per XCD, the default)."""
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
names = ["tx", "rx", "H", "H_ls", "H_mmse", "pilot_symbols", "pilot_mask"]
NSETS = 6
sets, pads = [], []
for i in range(NSETS):
    sets.append(eng.run_frames(d_dev, seed=1, outputs=names))
    pads.append(torch.empty((i + 1) * 123_456_789, dtype=torch.uint8, device=eng.device))
variants = [None, "32", "512", "s7", "s1249", "s4999"]       # None = default contiguous eighths; "sN" = scatter stride N; plus identity
res = {}
for it in range(6):
    for j, o in enumerate(sets):
        for v in variants + ["identity"]:
            os.environ.pop("CHEST_XCD_BLOCK", None); os.environ.pop("CHEST_NO_XCD_REMAP", None); os.environ.pop("CHEST_FRAME_STRIDE", None)
            if v is not None and v.startswith("s"):
                os.environ["CHEST_FRAME_STRIDE"] = v[1:]
            elif v == "identity":
                os.environ["CHEST_NO_XCD_REMAP"] = "1"
            elif v is not None:
                os.environ["CHEST_XCD_BLOCK"] = v
            eng.profile_begin(1)
            eng.run_frames(d_dev, seed=1, frame_offset=it * F, outputs=names, out=o, check=False)
            k_ms = eng.profile_read()[1]
            if it >= 1:
                res.setdefault((j, v), []).append(k_ms)
os.environ.pop("CHEST_XCD_BLOCK", None); os.environ.pop("CHEST_NO_XCD_REMAP", None); os.environ.pop("CHEST_FRAME_STRIDE", None)
print("frame_kernel ms; rows = output sets, columns = mapping")
print("set   " + "".join(f"{('eighths' if v is None else (v if v == 'identity' or v.startswith('s') else 'block ' + v)):>11s}" for v in variants + ["identity"]))
for j in range(NSETS):
    print(f"{j:3d}   " + "".join(f"{np.mean(res[(j, v)]):11.3f}" for v in variants + ["identity"]))
