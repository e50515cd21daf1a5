#!/usr/bin/env python3
"""Reference NMSE-vs-SNR curve (BASELINE config 3's shape: ETU 370.4 Hz, MMSE + LS, SNR -5..30 dB, 10 % pilots), run on the
REAL reference in the build container.  Reduced to 2x2 and 30 frames per SNR because the reference needs ~0.7 s per 2x2
frame (4x4: ~4.9 s).  Writes tests/golden/snr_sweep_ref.json: per SNR the frame-averaged MSE of LS and MMSE (against
H_true, as evaluate_estimator scores them) and the mean channel power."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/src")
import baseline_estimators as be
import channel_simulator as cs
cfg = {"ofdm": {"fft_size": 1024, "cp_length": 72, "num_symbols": 14, "useful_subcarriers": 600, "subcarrier_spacing": 15000.0},
       "mimo": {"num_tx_antennas": 2, "num_rx_antennas": 2}, "channel": {"carrier_freq": 2.0e9}}
np.random.seed(20260104)
out = {"channel": "ETU", "doppler_hz": 370.4, "pilot_density": 0.10, "frames_per_snr": 30, "mimo": "2x2", "snr": []}
for snr in (-5, 0, 5, 10, 15, 20, 25, 30):
    mls, mmm, pw = [], [], []
    for _ in range(30):
        sim = cs.simulate_transmission(cfg, "ETU", 370.4, snr, 0.10)
        pp = sim["pilot_pattern"]
        rx4 = np.repeat(sim["rx_symbols"][:, :, None, :], 2, axis=2)
        ls = be.LSEstimator("linear").estimate(rx4, sim["pilot_symbols"], pp.pilot_mask, pp.pilot_positions)
        mm = be.MMSEEstimator().estimate(rx4, sim["pilot_symbols"], pp.pilot_mask, pp.pilot_positions, snr)
        mls.append(be.evaluate_estimator(sim["channel"], ls)["mse"])
        mmm.append(be.evaluate_estimator(sim["channel"], mm)["mse"])
        pw.append(float(np.mean(np.abs(sim["channel"]) ** 2)))
    out["snr"].append({"snr_db": snr, "mse_ls": float(np.mean(mls)), "mse_mmse": float(np.mean(mmm)), "power": float(np.mean(pw)),
                       "mse_ls_std": float(np.std(mls)), "mse_mmse_std": float(np.std(mmm))})
    print(out["snr"][-1], flush=True)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "snr_sweep_ref.json"), "w"), indent=1)
