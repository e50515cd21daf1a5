#!/usr/bin/env python3
"""interp_from_pilots_tfold_kernel (config 5's interpolation + T-fold stores) under a chest_tune knob, same process, same buffers:
    python tools/ab_interp.py <knob> <value a> <value b>[,<value c>...] [frames=2304]        e.g.  interp_xcd -1 0   |   interp_mode 0 1,2,3"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
E = importlib.import_module("channel-estimation-in-5g-network_amd.engine")
knob, va, vbs = sys.argv[1], int(sys.argv[2]), [int(x) for x in sys.argv[3].split(",")]
F = int(sys.argv[4]) if len(sys.argv) > 4 else 2304
geom = E.Geometry(fft_size=2048, cp_length=144, useful_subcarriers=1200, num_tx=4, num_rx=4)
eng = E.Engine(geom, 0)
S, nsc, G, R, T = geom.num_symbols, geom.num_sc, geom.grid, 4, 4
P = int(G * 0.1)
slot = int(eng.build_patterns(0, [P], seed=1, pattern_ids=np.asarray([0], np.uint64), method="linear")[0])
scr = eng.mmse_full_scratch(slot, F * R)
scr.view(torch.float32).normal_()                  # (random pilot values: the data written is what the product writes)
out = eng.empty_c64(F, S, R, T, nsc)
nbytes = out.numel() * 8
if knob == "interp_mode":                            # every value of the knob that computes must give the same bits
    ref = None
    for v in [va] + vbs:
        if v in (2, 3):
            continue
        eng.tune(knob, v); out.zero_(); eng.mmse_full_interp_pilots(F, slot, scr, out); torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        else:
            print(f"{knob} {v} == {knob} {va}: {bool(torch.equal(torch.view_as_real(out), torch.view_as_real(ref)))}", flush=True)
    del ref
for rep in range(3):
    for v in [x for vb in vbs for x in (va, vb)]:
        eng.tune(knob, v)
        for _ in range(2):
            eng.mmse_full_interp_pilots(F, slot, scr, out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            eng.mmse_full_interp_pilots(F, slot, scr, out)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(f"{knob} {v:3d}: {ms:.3f} ms per {F} frames = {nbytes / ms / 1e9:.2f} TB/s of H_mmse = {nbytes / ms / 1e9 / 8:.3f} of 8 TB/s", flush=True)
# reference points on the SAME buffer: a plain fill (torch's vectorised store kernel) and a device-to-device copy of half of it
of = torch.view_as_real(out)
for name, fn, nb in (("torch fill_", lambda: of.fill_(1.0), nbytes),
                     ("torch copy_ (half -> half, read + write)", lambda: of[: F // 2].copy_(of[F // 2: 2 * (F // 2)]), nbytes // 2)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{name}: {ms:.3f} ms = {nb / ms / 1e9:.2f} TB/s written", flush=True)
