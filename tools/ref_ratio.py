#!/usr/bin/env python3
"""Calibration of bench.py's `cpu_baseline` (the oracle in faithful mode, timed on the GPU box's host cores) against the REFERENCE
itself: both run here, in the build container, on the same machine, one thread, the same frames.

    python tools/ref_ratio.py [--frames 6] [--reference /root/reference]  ->  profiles/cpu_ref_ratio.json

`ref_ratio` = seconds per frame of the reference (simulate_transmission + LSEstimator('linear').estimate + MMSEEstimator().estimate,
the loop body of src/dataset_generator.py:34-89 plus the MMSE label) / seconds per frame of the oracle's faithful mode on BASELINE
config 2 (2x2, 1024-FFT, EVA 111.1 Hz, 10 dB, 10 % pilots).  A ratio above 1 means the oracle is FASTER than the reference, i.e.
bench.py's cpu_baseline flatters the CPU by that factor; bench.py reports it beside the value (`cpu_baseline.ref_ratio`,
`cpu_baseline.reference_equivalent` = value / ref_ratio).  The reference is only imported here; nothing of it travels."""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "cpu_ref_ratio.json"))
    a = ap.parse_args()
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(1)
    except Exception:
        lim = None
    import bench
    sys.path.insert(0, os.path.join(a.reference, "src"))
    import baseline_estimators as be
    import channel_simulator as cs
    w = bench.WORKLOADS["cfg2_2x2_eva_ls_mmse"]
    cfg = {"ofdm": {"fft_size": w["N"], "cp_length": w["cp"], "num_symbols": 14, "useful_subcarriers": w["U"], "subcarrier_spacing": 15000.0},
           "mimo": {"num_tx_antennas": w["T"], "num_rx_antennas": w["R"]}, "channel": {"carrier_freq": 2.0e9}}
    T = w["T"]

    def ref_frames(n, seed):
        np.random.seed(seed)
        for _ in range(n):
            sim = cs.simulate_transmission(cfg, w["ct"], w["fd"], w["snr"], w["dens"])
            pp = sim["pilot_pattern"]
            rx4 = np.repeat(sim["rx_symbols"][:, :, None, :], T, axis=2)
            ls = be.LSEstimator("linear").estimate(rx4, sim["pilot_symbols"], pp.pilot_mask, pp.pilot_positions)
            mm = be.MMSEEstimator().estimate(rx4, sim["pilot_symbols"], pp.pilot_mask, pp.pilot_positions, w["snr"])
            be.evaluate_estimator(sim["channel"], ls); be.evaluate_estimator(sim["channel"], mm)

    ref_frames(1, 1); bench._cpu_frames(("cfg2_2x2_eva_ls_mmse", 1, 1))           # warm both (imports, Qhull, FFT plans)
    rounds = []
    for rep in range(3):                                                           # alternate, keep the best of each
        t0 = time.perf_counter(); ref_frames(a.frames, 100 + rep); t_ref = (time.perf_counter() - t0) / a.frames
        t0 = time.perf_counter(); bench._cpu_frames(("cfg2_2x2_eva_ls_mmse", a.frames, 200 + rep)); t_or = (time.perf_counter() - t0) / a.frames
        rounds.append((t_ref, t_or))
    t_ref, t_or = min(r for r, _ in rounds), min(o for _, o in rounds)
    rec = {"workload": "cfg2_2x2_eva_ls_mmse", "frames_per_repetition": a.frames, "repetitions": len(rounds), "threads": 1,
           "reference_s_per_frame": t_ref, "oracle_faithful_s_per_frame": t_or, "ref_ratio": t_ref / t_or,
           "all_repetitions": rounds, "machine": platform.processor() or platform.machine(), "cpu_count": os.cpu_count(),
           "numpy": np.__version__, "note": "measured in the build container (the reference never travels to the GPU box); ref_ratio > 1: "
                                            "the oracle is faster than the reference by that factor"}
    json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(rec))
    del lim


if __name__ == "__main__":
    main()
