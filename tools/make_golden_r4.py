#!/usr/bin/env python3
"""Round-4 additions to the golden fixtures, generated from the REAL reference in the build container (earlier fixtures stay
untouched):

    python tools/make_golden_r4.py --reference /root/reference --out tests/golden

  full_cfg5_4x4_2048_etu.npz     BASELINE config 5's geometry through the reference itself: 4x4, 2048-FFT (cp 144, 1 200 used
                                 subcarriers -> nsc 1 199), ETU, 555.6 Hz (300 km/h @ 2 GHz), 10 dB, 10 % pilots (P = 1 678):
                                 simulate_transmission + LSEstimator('linear') + MMSEEstimator() (default statistics), same
                                 layout as tools/make_golden.py's full_* cases (complex64 tensors + float64 checksums), the
                                 oracle asserted <= 1e-12 first.
  fullcov_cfg5_4x4_2048_etu.npz  the same frame through MMSEEstimator(channel_covariance=R, estimate_statistics=False).estimate
                                 (src/baseline_estimators.py:155-196, :232-270) -- sixteen 1 678 x 1 678 complex inverses on the
                                 reference -- with the harness covariance of bench.fullcov_covariance (checksum stored).  Inputs
                                 are those of full_cfg5_4x4_2048_etu.npz (same seed).
  ml_packing.npz                 the training-side feature packing of SURVEY 8f-2 through the reference's own code:
                                 dataset_generator.prepare_ml_inputs (both `normalize` modes, src/dataset_generator.py:183-227),
                                 train.ChannelDataset (dataset-level statistics + __getitem__, src/train.py:22-94) and
                                 run_phase4_training.ChannelEstimationDataset (per-tensor std, run_phase4_training.py:60-112) on eight
                                 reduced-grid 2x2 frames produced by the reference's simulate_transmission + LSEstimator.

Fixtures hold data only (inputs, expected outputs, versions); no reference source text.
"""
import argparse
import os
import sys
import types

import numpy as np
import scipy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import cpu_ref as O  # noqa: E402
import bench  # noqa: E402
import make_golden as MG  # noqa: E402

CFG5 = dict(N=2048, cp=144, U=1200, T=4, R=4, ct="ETU", fd=555.6, snr=10, dens=0.10, seed=5)


def cfg5(cs, be, out_dir):
    c = CFG5
    name = "full_cfg5_4x4_2048_etu"
    MG.run_case((cs, be), name, c["seed"], c["N"], c["cp"], c["U"], c["T"], c["R"], c["ct"], c["fd"], c["snr"], c["dens"], out_dir,
                store_c64=True, with_nearest=False)
    # the same frame again (same seed -> same draws), through the full-covariance estimator of the reference
    cfg = MG.make_cfg(c["N"], c["cp"], c["U"], c["T"], c["R"])
    np.random.seed(c["seed"])
    sim = cs.simulate_transmission(cfg, c["ct"], c["fd"], c["snr"], c["dens"])
    pp = sim["pilot_pattern"]
    T = c["T"]
    rx4 = np.repeat(sim["rx_symbols"][:, :, None, :], T, axis=2)
    cov = bench.fullcov_covariance(c["N"], c["cp"], c["U"], T, c["ct"], c["fd"], pp.pilot_indices)
    H_fc = be.MMSEEstimator(channel_covariance=cov, estimate_statistics=False).estimate(
        rx4, sim["pilot_symbols"], pp.pilot_mask, pp.pilot_positions, c["snr"])
    with np.load(os.path.join(out_dir, name + ".npz")) as g:
        assert np.array_equal(g["pilot_indices"], pp.pilot_indices)
        assert MG.rel(g["rx"], sim["rx_symbols"]) < 2e-7
    o_fc = O.mmse_estimate(rx4, sim["pilot_symbols"], pp.pilot_mask, pp.pilot_positions, c["snr"], covariance=cov,
                           estimate_statistics=False)
    e = MG.rel(o_fc, H_fc)
    assert e < 1e-9, e
    for t in range(1, T):
        assert np.array_equal(H_fc[:, :, t, :], H_fc[:, :, 0, :])
    m = be.evaluate_estimator(sim["channel"], H_fc)
    path = os.path.join(out_dir, "fullcov_cfg5_4x4_2048_etu.npz")
    np.savez_compressed(path, inputs_from=name, seed=c["seed"], snr_db=float(c["snr"]), doppler_hz=c["fd"], numpy=np.__version__,
                        scipy=scipy.__version__, H_fullcov=H_fc[:, :, 0, :].astype(np.complex64), sum_H_fullcov=np.sum(H_fc),
                        mse_fullcov=m["mse"], nmse_db_fullcov=m["nmse_db"],
                        cov_checksum=np.array([np.sum(cov), np.sum(np.abs(cov) ** 2)]))
    print(f"fullcov_cfg5_4x4_2048_etu          ok  oracle vs reference {e:.2e}  {os.path.getsize(path) / 1024:.0f} KiB")


def ml_packing(ref_root, cs, be, out_dir, frames=8, seed=404):
    """Eight reduced-grid frames (U = 72 -> 14 x 71) in the stacked layout of run_phase3_dataset_generation.py:135-143, packed by
    the three pieces of reference code that turn dataset records into network inputs."""
    import torch
    if "h5py" not in sys.modules:                      # dataset_generator imports h5py at module level (only _save_h5 uses it)
        try:
            import h5py  # noqa: F401
        except ImportError:
            sys.modules["h5py"] = types.ModuleType("h5py")
    if "torch.utils.tensorboard" not in sys.modules:   # train.py imports SummaryWriter at module level (only Trainer uses it);
        try:                                           # tensorboard is not installed here: an empty stand-in for the import
            import torch.utils.tensorboard  # noqa: F401
        except ImportError:
            stub = types.ModuleType("torch.utils.tensorboard")
            stub.SummaryWriter = object
            sys.modules["torch.utils.tensorboard"] = stub
    import dataset_generator as dg
    import train as tr
    sys.path.insert(0, ref_root)
    import run_phase4_training as p4
    N, cp, U, T, R = 1024, 72, 72, 2, 2
    cfg = MG.make_cfg(N, cp, U, T, R)
    np.random.seed(seed)
    params = [("EPA", 10.0, 0, 0.05), ("EVA", 50.0, 10, 0.10), ("ETU", 100.0, 20, 0.10), ("EVA", 200.0, 30, 0.05),
              ("ETU", 10.0, -5, 0.10), ("EPA", 100.0, 15, 0.05), ("EVA", 111.1, 5, 0.10), ("ETU", 370.4, 25, 0.05)][:frames]
    recs = []
    for ct, fd, snr, dens in params:
        sim = cs.simulate_transmission(cfg, ct, fd, snr, dens)
        pp = sim["pilot_pattern"]
        rx4 = np.repeat(sim["rx_symbols"][:, :, None, :], T, axis=2)
        ls = be.LSEstimator("linear").estimate(rx4, sim["pilot_symbols"], pp.pilot_mask, pp.pilot_positions)
        recs.append({"rx_symbols": sim["rx_symbols"].astype(np.complex64), "tx_symbols": sim["tx_symbols"].astype(np.complex64),
                     "pilot_mask": pp.pilot_mask.astype(np.float32), "H_ls": ls.astype(np.complex64),
                     "H_true": sim["channel"].astype(np.complex64), "snr_db": float(snr), "doppler_hz": float(fd),
                     "channel_type": ct})
    stack = {k: np.stack([r[k] for r in recs]) for k in ("rx_symbols", "pilot_mask", "H_ls", "H_true")}
    # (a) dataset_generator.prepare_ml_inputs, both modes -- per sample, channels-last
    pm = [dg.prepare_ml_inputs(r, normalize=True) for r in recs]
    pr = [dg.prepare_ml_inputs(r, normalize=False) for r in recs]
    # (b) train.ChannelDataset: dataset-level statistics, channels-first
    path = os.path.join(out_dir, "_ml_tmp.npz")
    np.savez(path, snr_db=np.array([r["snr_db"] for r in recs], np.float32), **stack)
    try:
        ds = tr.ChannelDataset(path, normalize=True)
        items = [ds[i] for i in range(frames)]
        ds_raw = tr.ChannelDataset(path, normalize=False)
        items_raw = [ds_raw[i] for i in range(frames)]
        # (c) run_phase4_training.ChannelDataset: per-tensor std, channels-first
        ds4 = p4.ChannelEstimationDataset(path, normalize=True)
        items4 = [ds4[i] for i in range(frames)]
    finally:
        os.remove(path)
    payload = dict(
        fft_size=N, cp_length=cp, useful_subcarriers=U, num_symbols=14, num_tx=T, num_rx=R, seed=seed, numpy=np.__version__,
        torch=torch.__version__, **stack,
        prep_norm_inputs=np.stack([a for a, _ in pm]).astype(np.float32), prep_norm_targets=np.stack([b for _, b in pm]).astype(np.float32),
        prep_raw_inputs=np.stack([a for a, _ in pr]).astype(np.float32), prep_raw_targets=np.stack([b for _, b in pr]).astype(np.float32),
        train_inputs=np.stack([a.numpy() for a, _, _ in items]), train_targets=np.stack([b.numpy() for _, b, _ in items]),
        train_raw_inputs=np.stack([a.numpy() for a, _, _ in items_raw]), train_raw_targets=np.stack([b.numpy() for _, b, _ in items_raw]),
        train_stats=np.array([ds.rx_mean, ds.rx_std, ds.H_ls_mean, ds.H_ls_std, ds.H_true_mean, ds.H_true_std], dtype=np.float64),
        p4_stats=np.array([ds4.rx_std, ds4.H_ls_std, ds4.H_true_std], dtype=np.float64),
        p4_inputs=np.stack([a.numpy() for a, _, _ in items4]), p4_targets=np.stack([b.numpy() for _, b, _ in items4]),
    )
    out = os.path.join(out_dir, "ml_packing.npz")
    np.savez_compressed(out, **payload)
    print(f"ml_packing                         ok  {frames} frames  {os.path.getsize(out) / 1024:.0f} KiB")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default="", help="cfg5 | ml")
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(args.reference, "src"))
    import baseline_estimators as be
    import channel_simulator as cs
    if args.only in ("", "cfg5"):
        cfg5(cs, be, args.out)
    if args.only in ("", "ml"):
        ml_packing(args.reference, cs, be, args.out)


if __name__ == "__main__":
    main()
