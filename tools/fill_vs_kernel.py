#!/usr/bin/env python3
"""The frame kernel's write rate beside a plain fill of the SAME output allocations, in one process (the pool's boxes and a
process's allocations differ in their write level, DESIGN 5 K2: only a same-process comparison says how far the kernel's store
pattern is from a purely sequential one).   python tools/fill_vs_kernel.py [frames=10000] [shape=cfg2|siso|cfg3]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
E = importlib.import_module("channel-estimation-in-5g-network_amd.engine")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
shape = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
geom = {"cfg2": E.Geometry(), "siso": E.Geometry(num_tx=1, num_rx=1), "cfg3": E.Geometry(num_tx=4, num_rx=4)}[shape]
ct, fd = {"cfg2": ("EVA", 111.1), "siso": ("EPA", 5.56), "cfg3": ("ETU", 370.4)}[shape]
eng = E.Engine(geom, 0)
G = geom.grid
P = int(G * 0.1)
slots = eng.build_patterns(0, np.full(10, P, np.int32), seed=2024, pattern_ids=np.arange(10, dtype=np.uint64), method="linear")
descs = E.make_descs(ct, fd, 10.0, np.asarray(slots, np.int32)[np.arange(F) // ((F + 9) // 10)], n=F)
d_dev = torch.from_numpy(descs.view(np.uint8).copy()).to(eng.device)
names = ["tx", "rx", "H", "H_ls", "H_mmse", "pilot_symbols", "pilot_mask"]


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for rep in range(3):
    out = eng.run_frames(d_dev, seed=1, outputs=names)                    # a fresh allocation of the seven tensors per repetition
    nbytes = sum(out[k].numel() * out[k].element_size() for k in names)
    eng.profile_begin(12)
    ms_step = timed(lambda: eng.run_frames(d_dev, seed=1, outputs=names, out=out, check=False))
    ms_k = eng.profile_read()[1]
    eng.profile_begin(0)
    views = [torch.view_as_real(out[k]) if out[k].is_complex() else out[k] for k in names]
    rnd = [torch.randn_like(v, dtype=torch.float32) if v.dtype == torch.float32 else None for v in views]
    ms_fill = timed(lambda: [v.fill_(1) for v in views])
    ms_copy = timed(lambda: [v.copy_(r) for v, r in zip(views, rnd) if r is not None], n=5)
    cb = sum(v.numel() * 4 for v, r in zip(views, rnd) if r is not None)
    del rnd
    print(f"allocation {rep}: frame kernel {ms_k:.3f} ms = {nbytes / ms_k / 1e9:.2f} TB/s (step {ms_step:.3f}) | fill_ of the same tensors {ms_fill:.3f} ms = "
          f"{nbytes / ms_fill / 1e9:.2f} TB/s | copy_ of random data into them {ms_copy:.3f} ms = {cb / ms_copy / 1e9:.2f} TB/s written (+ as much read)", flush=True)
    del out, views
