#!/usr/bin/env python3
"""In-process ablation of the frame kernel by requested outputs: one process, one GPU, the variants interleaved
round-robin, kernel time from the library's HIP events.  Shows where the step time goes (phase W = tx/rx/H synthesis
and stores, phase E = estimate interpolation and stores) at equal clocks.

    python tools/ablate_inprocess.py [--rounds 12] [--frames 10000]
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "channel-estimation-in-5g-network_amd"
BYTES = {"tx": 8 * 14 * 599 * 2, "rx": 8 * 14 * 599 * 2, "H": 8 * 14 * 599 * 4, "H_ls": 8 * 14 * 599 * 4, "H_mmse": 8 * 14 * 599 * 4,
         "pilot_symbols": 8 * 838, "pilot_mask": 14 * 599}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--frames", type=int, default=10000)
    args = ap.parse_args()
    import torch
    E = importlib.import_module(PKG + ".engine")
    geom = E.Geometry()
    eng = E.Engine(geom, 0)
    G = geom.grid
    P = int(G * 0.1)
    F = args.frames
    slots = []
    for gidx in range((F + 1023) // 1024):
        perm = np.arange(G)
        np.random.RandomState(2024 + gidx).shuffle(perm)
        slots.append(eng.pattern_slot(np.sort(perm[:P]), "linear"))
    descs = E.make_descs("EVA", 111.1, 10.0, np.asarray(slots, dtype=np.int32)[np.arange(F) // 1024], n=F)
    d_dev = torch.from_numpy(descs.view(np.uint8).copy()).to(eng.device)
    full = ["tx", "rx", "H", "H_ls", "H_mmse", "pilot_symbols", "pilot_mask"]
    variants = {
        "all seven outputs": full,
        "tensors only (no pilot_symbols / pilot_mask)": ["tx", "rx", "H", "H_ls", "H_mmse"],
        "phase W only: tx rx H": ["tx", "rx", "H"],
        "tx rx": ["tx", "rx"],
        "H only": ["H"],
        "rx only": ["rx"],
        "estimates only: H_ls H_mmse": ["H_ls", "H_mmse"],
        "H_ls only": ["H_ls"],
        "rx + estimates": ["rx", "H_ls", "H_mmse"],
        "metrics only (no tensor stores)": ["metrics"],
    }
    # ONE set of output tensors, placed once (Engine.place_outputs) and shared by every variant: the time of this kernel
    # depends on where its outputs live, so separate allocations per variant would confound the comparison.
    placed, rep = eng.place_outputs(d_dev, full + ["metrics"], seed=1, tries=4)
    print("placement:", rep)
    outs = {k: {n: placed[n] for n in v} for k, v in variants.items()}
    t = {k: [] for k in variants}
    for i in range(args.rounds + 2):
        for k, v in variants.items():
            eng.profile_begin(1)
            eng.run_frames(d_dev, seed=1, frame_offset=i * F, outputs=v, out=outs[k], check=False)
            _, k_ms, _ = eng.profile_read()
            if i >= 2:
                t[k].append(k_ms)
    for k, v in variants.items():
        a = np.array(t[k])
        nb = sum(BYTES.get(n, 0) for n in v) * F
        print(f"{k:48s} {a.mean():7.3f} ms (sd {a.std():.3f})  {nb / 1e9:6.2f} GB  {nb / a.mean() * 1e-9:8.2f} TB/s" if nb else
              f"{k:48s} {a.mean():7.3f} ms (sd {a.std():.3f})")


if __name__ == "__main__":
    main()
