#!/usr/bin/env python3
"""Where a frame-kernel unit spends its cycles: runs the DIAGNOSTIC library (make -C <pkg>/csrc stamps -> libchest_stamps.so,
in-kernel s_memtime stamps around compute / barriers / slab writes / store-out) on the config-2 workload and prints each
segment's share of a wave's lifetime.  Read the shares, not the run time (the stamps' waits forbid overlaps the real
kernel has).  CHEST_LIB selects the library, so this must be set before the package is imported."""
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "channel-estimation-in-5g-network_amd"
os.environ["CHEST_LIB"] = os.path.join(ROOT, PKG, os.environ.get("STAMPS_LIB", "libchest_stamps.so"))
NAMES = ["W compute", "W barrier B", "W slab write", "W barrier A", "W store-out", "barrier after W", "pilots+alpha",
         "E compute", "E barrier B", "E slab write", "E barrier A", "E store-out"]


def main():
    import torch
    E = importlib.import_module(PKG + ".engine")
    N = importlib.import_module(PKG + "._native")
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    cfg5 = os.environ.get("STAMPS_SHAPE") == "cfg5"          # the 2048-FFT 4x4 launch of config 5 (no estimates; 10 compute waves + store waves)
    siso = os.environ.get("STAMPS_SHAPE") == "siso"          # config 1's shape: 1x1, EPA, 5.56 Hz (two symbols per unit)
    cfg3 = os.environ.get("STAMPS_SHAPE") == "cfg3"          # config 3's shape: 4x4, 1024-FFT, ETU (5 compute + 2 store waves, two workgroups per CU)
    geom = E.Geometry(fft_size=2048, cp_length=144, useful_subcarriers=1200, num_tx=4, num_rx=4) if cfg5 else \
        (E.Geometry(num_tx=1, num_rx=1) if siso else (E.Geometry(num_tx=4, num_rx=4) if cfg3 else E.Geometry()))
    eng = E.Engine(geom, 0)
    if os.environ.get("STAMPS_ONE_SLAB"):
        eng.tune("frame_two_slabs", -1)
    eng.tune("frame_min_waves", int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    G = geom.grid
    P = int(G * 0.1)
    per_frame = len(sys.argv) > 3 and sys.argv[3] == "pf"           # one device-built pattern (and plan) per frame
    if per_frame:
        sl = eng.build_patterns(0, np.full(F, P, np.int32), seed=2024, pattern_ids=np.arange(F, dtype=np.uint64), method=os.environ.get("STAMPS_PLAN", "linear_compact"))
        descs = E.make_descs("EVA", 111.1, 10.0, np.asarray(sl, np.int32), n=F)
    else:
        slots = []
        for g in range((F + 1023) // 1024):
            perm = np.arange(G); np.random.RandomState(2024 + g).shuffle(perm)
            slots.append(eng.pattern_slot(np.sort(perm[:P]), "linear"))
        descs = E.make_descs("ETU" if (cfg5 or cfg3) else ("EPA" if siso else "EVA"), 555.6 if cfg5 else (370.4 if cfg3 else (5.56 if siso else 111.1)), 10.0, np.asarray(slots, np.int32)[np.arange(F) // 1024], n=F)
    d_dev = torch.from_numpy(descs.view(np.uint8).copy()).to(eng.device)
    names = ["tx", "rx", "H", "pilot_symbols", "pilot_mask", "h_ls_pilots"] if cfg5 else ["tx", "rx", "H", "H_ls", "H_mmse", "pilot_symbols", "pilot_mask"]
    out = eng.run_frames(d_dev, seed=1, outputs=names)
    buf = torch.zeros(F * 16 * 16, dtype=torch.int64, device=eng.device)
    lib = N.lib()
    lib.chest_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    assert lib.chest_debug_set_stamp_buffer(buf.data_ptr()) == 0
    for _ in range(3):
        eng.profile_begin(1)
        eng.run_frames(d_dev, seed=1, outputs=names, out=out, check=False)
        ms = eng.profile_read()[1]
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(F, 16, 16)
    # workgroup lifetimes (s_memrealtime, 100 MHz), in-kernel clock, residency
    rt0, rt1, mt = raw[:, 0, 12].astype(np.float64), raw[:, 0, 13].astype(np.float64), raw[:, 0, 14].astype(np.float64)
    life_us = (rt1 - rt0) / 100.0
    print(f"workgroup lifetime {np.median(life_us):.1f} us (p10 {np.percentile(life_us, 10):.1f}, p90 {np.percentile(life_us, 90):.1f}); "
          f"in-kernel clock {np.median(mt / np.maximum(rt1 - rt0, 1)) * 100:.0f} MHz; kernel span {(rt1.max() - rt0.min()) / 100:.0f} us; "
          f"average resident workgroups {life_us.sum() / ((rt1.max() - rt0.min()) / 100):.0f} (of 256 CUs)")
    hw = raw[:, 0, 15]
    cu_key = np.asarray([((int(h) >> 32) << 8) | ((int(h) >> 8) & 0xFF) for h in hw])      # (xcc, se/sh/cu id)
    conc = []
    for key in np.unique(cu_key):
        m = cu_key == key
        ev = sorted([(t, 1) for t in rt0[m]] + [(t, -1) for t in rt1[m]])
        cur = peak = 0; area = 0.0; last = ev[0][0]
        for t, d in ev:
            area += cur * (t - last); last = t
            cur += d; peak = max(peak, cur)
        conc.append((m.sum(), peak, area / max(ev[-1][0] - ev[0][0], 1)))
    conc = np.asarray(conc)
    # how long a freed slot stays empty: the k-th workgroup start on a CU (beyond the first `peak`) against the (k - peak)-th end
    gaps = []
    for key in np.unique(cu_key):
        m = cu_key == key
        st_, en_ = np.sort(rt0[m]), np.sort(rt1[m])
        pk = int(conc[list(np.unique(cu_key)).index(key), 1])
        if len(st_) > pk:
            gaps.extend(((st_[pk:] - en_[: len(st_) - pk]) / 100.0).tolist())
    gaps = np.asarray(gaps)
    if os.environ.get("STAMPS_TIMELINE"):                     # the workgroups of the first CUs, in start order: frame, start, end (us since the launch's first start)
        t0 = rt0.min()
        for key in np.unique(cu_key)[: int(os.environ["STAMPS_TIMELINE"])]:
            idx = np.nonzero(cu_key == key)[0]
            idx = idx[np.argsort(rt0[idx])]
            print(f"CU {key:#x}: " + "  ".join(f"{i}:{(rt0[i] - t0) / 100:.0f}-{(rt1[i] - t0) / 100:.0f}" for i in idx))
    if len(gaps):
        print(f"slot refill gap (end of a workgroup -> start of the next one on that CU): median {np.median(gaps):.1f} us, p10 {np.percentile(gaps, 10):.1f}, "
              f"p90 {np.percentile(gaps, 90):.1f}, mean {gaps.mean():.1f} us  ({len(gaps)} refills)")
    print(f"CUs seen {len(conc)}; frames per CU {conc[:, 0].min():.0f}..{conc[:, 0].max():.0f}; peak resident workgroups per CU "
          f"{conc[:, 1].min():.0f}..{conc[:, 1].max():.0f}; time-averaged {conc[:, 2].mean():.2f}")
    nw = 14 if cfg5 else (7 if cfg3 else 5)
    st = raw[:, :nw, :12].astype(np.float64)      # waves per workgroup (compute waves first, then any store waves)
    tot = st.sum(axis=2)
    print(f"stamped kernel {ms:.3f} ms; mean stamped cycles per wave per frame {tot.mean():,.0f}")
    for i, n in enumerate(NAMES):
        print(f"  {n:16s} {st[:, :, i].mean():10,.0f} cycles  {st[:, :, i].sum() / tot.sum():6.1%}   per wave: " +
              " ".join(f"{st[:, w, i].mean():8,.0f}" for w in range(nw)))


if __name__ == "__main__":
    main()
