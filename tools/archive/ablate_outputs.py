#!/usr/bin/env python3
"""Frame-kernel time by requested output subset, one process, one set of buffers (first allocation):
which part of the step is synthesis, which is stores.  `--rg` sets the slab rx group, `--min-waves 4` the 128-VGPR instantiation."""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "channel-estimation-in-5g-network_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10000)
    ap.add_argument("--rg", type=int, default=0)
    ap.add_argument("--min-waves", type=int, default=0)
    ap.add_argument("--mimo", type=int, default=2)
    ap.add_argument("--reps", type=int, default=7)
    args = ap.parse_args()
    import torch
    E = importlib.import_module(PKG + ".engine")
    geom = E.Geometry(num_tx=args.mimo, num_rx=args.mimo)
    eng = E.Engine(geom, 0)
    eng.tune("frame_rx_group", args.rg)
    eng.tune("frame_min_waves", args.min_waves)
    G = geom.grid
    P = int(G * 0.1)
    F = args.frames
    slots = []
    for g in range((F + 1023) // 1024):
        perm = np.arange(G); np.random.RandomState(2024 + g).shuffle(perm)
        slots.append(eng.pattern_slot(np.sort(perm[:P]), "linear"))
    descs = E.make_descs("EVA", 111.1, 10.0, np.asarray(slots, np.int32)[np.arange(F) // 1024], n=F)
    d_dev = torch.from_numpy(descs.view(np.uint8).copy()).to(eng.device)
    full = ["tx", "rx", "H", "H_ls", "H_mmse", "pilot_symbols", "pilot_mask"]
    out = eng.run_frames(d_dev, seed=1, outputs=full + ["metrics"])
    subsets = [("all seven", full), ("tx rx H", ["tx", "rx", "H"]), ("H_ls H_mmse", ["H_ls", "H_mmse"]), ("H only", ["H"]),
               ("rx only", ["rx"]), ("H_ls only", ["H_ls"]), ("metrics only", ["metrics"]), ("tx only", ["tx"])]
    T = R = args.mimo
    el = {"tx": T, "rx": R, "H": R * T, "H_ls": R * T, "H_mmse": R * T}
    for name, names in subsets:
        ts = []
        for _ in range(args.reps):
            eng.profile_begin(1)
            eng.run_frames(d_dev, seed=1, outputs=names, out={k: out[k] for k in names}, check=False)
            ts.append(eng.profile_read()[1])
        ms = float(np.median(ts[1:]))
        gb = sum(el.get(n, 0) for n in names) * 8 * G * F / 1e9
        print(f"{name:14s} {ms:7.3f} ms   {gb:6.2f} GB  -> {gb / ms if ms else 0:5.2f} TB/s", flush=True)
    eng.check_status()


if __name__ == "__main__":
    main()
