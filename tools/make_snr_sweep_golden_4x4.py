#!/usr/bin/env python3
"""Reference NMSE-vs-SNR curve at BASELINE config 3's FULL shape (4x4, ETU 370.4 Hz, LS + MMSE, SNR -5..30 dB, 10 % pilots),
run on the REAL reference in the build container: one process per SNR point (8 processes), FRAMES frames each
(the reference needs ~5 s per 4x4 frame).  Follows test_phase2_snr_sweep.py:18-64,74-96 of the reference: per frame
simulate_transmission -> rx repeated over tx -> LSEstimator('linear').estimate / MMSEEstimator().estimate ->
evaluate_estimator against H_true.  Writes tests/golden/snr_sweep_ref_4x4.json."""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAMES = int(os.environ.get("SWEEP_FRAMES", "50"))
SNRS = (-5, 0, 5, 10, 15, 20, 25, 30)
CFG = {"ofdm": {"fft_size": 1024, "cp_length": 72, "num_symbols": 14, "useful_subcarriers": 600, "subcarrier_spacing": 15000.0},
       "mimo": {"num_tx_antennas": 4, "num_rx_antennas": 4}, "channel": {"carrier_freq": 2.0e9}}


def one_snr(args):
    i, snr = args
    os.environ["OMP_NUM_THREADS"] = "1"
    sys.path.insert(0, "/root/reference/src")
    import baseline_estimators as be
    import channel_simulator as cs
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(1)
    except Exception:
        lim = None
    np.random.seed(20260104 + i)
    mls, mmm, pw = [], [], []
    for f in range(FRAMES):
        sim = cs.simulate_transmission(CFG, "ETU", 370.4, snr, 0.10)
        pp = sim["pilot_pattern"]
        rx4 = np.repeat(sim["rx_symbols"][:, :, None, :], 4, axis=2)
        ls = be.LSEstimator("linear").estimate(rx4, sim["pilot_symbols"], pp.pilot_mask, pp.pilot_positions)
        mm = be.MMSEEstimator().estimate(rx4, sim["pilot_symbols"], pp.pilot_mask, pp.pilot_positions, snr)
        mls.append(be.evaluate_estimator(sim["channel"], ls)["mse"])
        mmm.append(be.evaluate_estimator(sim["channel"], mm)["mse"])
        pw.append(float(np.mean(np.abs(sim["channel"]) ** 2)))
        if f % 10 == 9:
            print(f"snr {snr}: {f + 1}/{FRAMES}", flush=True)
    del lim
    return {"snr_db": snr, "seed": 20260104 + i, "mse_ls": float(np.mean(mls)), "mse_mmse": float(np.mean(mmm)),
            "power": float(np.mean(pw)), "mse_ls_std": float(np.std(mls)), "mse_mmse_std": float(np.std(mmm))}


if __name__ == "__main__":
    with mp.get_context("spawn").Pool(len(SNRS)) as pool:
        rows = pool.map(one_snr, list(enumerate(SNRS)))
    out = {"channel": "ETU", "doppler_hz": 370.4, "pilot_density": 0.10, "frames_per_snr": FRAMES, "mimo": "4x4",
           "numpy": np.__version__, "snr": rows}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "snr_sweep_ref_4x4.json"), "w"), indent=1)
    for r in rows:
        print(r)
