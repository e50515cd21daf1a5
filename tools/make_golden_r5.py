#!/usr/bin/env python3
"""Round-5 golden fixtures from the REAL reference (build container only; the reference never travels):

    python tools/make_golden_r5.py --reference /root/reference --out tests/golden

  r5_any_grid.npz          the estimator-side functions on grids no OFDM numerology produces -- (S, nsc) = (14, 64) 2x2 (EVEN
                           subcarrier count) and (7, 600) 1x2 -- with ARBITRARY rx_4d (not the tx-repeated one): LSEstimator('linear' /
                           'nearest').estimate, MMSEEstimator().estimate, MIMOChannel.apply_channel (seeded noise, replayed),
                           equalize_channel 'zf' / 'mmse' (src/baseline_estimators.py:83-117, :232-312; src/channel_simulator.py:313-345)
  r5_custom_profiles.npz   simulate_transmission with an EDITED ChannelModel.CHANNEL_PROFILES['EPA'] entry and an ADDED 'TDLX' entry whose
                           first path is not at delay 0 and two of whose paths collide on one sample delay (the class dict is read at
                           src/channel_simulator.py:72), draws replayed, + generate_time_varying_channel of the added profile

The oracle restatement is asserted <= 1e-12 against every reference output before anything is written.  Data only."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpu_ref as O  # noqa: E402

TOL = 1e-12


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


def any_grid(cs, be, out):
    payload = {}
    for tag, (S, R, T, nsc, dens, snr, seed) in {"g64": (14, 2, 2, 64, 0.12, 12.0, 501), "g600": (7, 2, 1, 600, 0.05, 3.0, 502)}.items():
        np.random.seed(seed)
        pp = cs.PilotPattern(nsc, S, dens)
        P = len(pp.pilot_indices)
        rs = np.random.RandomState(seed + 1)
        rx4 = rs.randn(S, R, T, nsc) + 1j * rs.randn(S, R, T, nsc)
        xp = np.exp(2j * np.pi * rs.rand(P))
        ls = be.LSEstimator("linear").estimate(rx4, xp, pp.pilot_mask, pp.pilot_positions)
        nn = be.LSEstimator("nearest").estimate(rx4, xp, pp.pilot_mask, pp.pilot_positions)
        mm = be.MMSEEstimator().estimate(rx4, xp, pp.pilot_mask, pp.pilot_positions, snr)
        pos = (np.asarray(pp.pilot_positions[0]), np.asarray(pp.pilot_positions[1]))
        assert rel(O.ls_estimate(rx4, xp, pp.pilot_mask, pos, "linear"), ls) <= TOL
        assert rel(O.ls_estimate(rx4, xp, pp.pilot_mask, pos, "nearest"), nn) <= TOL
        assert rel(O.mmse_estimate(rx4, xp, pp.pilot_mask, pos, snr), mm) <= TOL
        # apply_channel: tx (S,T,nsc), H (S,R,T,nsc); the noise blocks are the next 2 * S * R * nsc normals of the global stream
        tx = np.exp(2j * np.pi * rs.rand(S, T, nsc))
        H = (rs.randn(S, R, T, nsc) + 1j * rs.randn(S, R, T, nsc)) / np.sqrt(2)
        ch = cs.MIMOChannel(cs.OFDMConfig(), cs.MIMOConfig(num_tx=T, num_rx=R), cs.ChannelModel("EPA", 10.0, 2e9, 15.36e6))
        np.random.seed(seed + 2)
        rx = ch.apply_channel(tx, H, snr)
        np.random.seed(seed + 2)
        n_re = np.random.randn(S, R, nsc); n_im = np.random.randn(S, R, nsc)
        assert rel(O.apply_channel(tx, H, snr, n_re, n_im), rx) <= TOL
        eq_zf = be.equalize_channel(rx, H, "zf")
        eq_mm = be.equalize_channel(rx, H, "mmse")
        assert rel(O.equalize(rx, H, "zf"), eq_zf) <= 1e-9 and rel(O.equalize(rx, H, "mmse"), eq_mm) <= 1e-9
        ev = be.evaluate_estimator(H, ls)
        payload.update({f"{tag}_shape": np.asarray([S, R, T, nsc]), f"{tag}_snr": snr, f"{tag}_pilot_indices": pp.pilot_indices.astype(np.int32),
                        f"{tag}_rx4": rx4, f"{tag}_xp": xp, f"{tag}_H_ls": ls, f"{tag}_H_nearest": nn, f"{tag}_H_mmse": mm,
                        f"{tag}_tx": tx, f"{tag}_H": H, f"{tag}_n_re": n_re, f"{tag}_n_im": n_im, f"{tag}_rx": rx,
                        f"{tag}_eq_zf": eq_zf, f"{tag}_eq_mmse": eq_mm, f"{tag}_mse": ev["mse"], f"{tag}_nmse_db": ev["nmse_db"]})
    np.savez_compressed(os.path.join(out, "r5_any_grid.npz"), numpy_version=np.__version__, **payload)
    print("r5_any_grid.npz written")


def custom_profiles(cs, be, out):
    saved = {k: dict(v) for k, v in cs.ChannelModel.CHANNEL_PROFILES.items()}
    o_saved = dict(O.PROFILES)
    tables = {
        # EPA edited in place: one delay moved, one power changed
        "EPA": (np.array([0, 30, 70, 90, 110, 250, 410]) * 1e-9, np.array([0.0, -1.0, -2.0, -3.0, -6.0, -17.2, -20.8])),
        # a new name: six paths, the first NOT at delay 0, paths 2 and 3 round to the same sample at 15.36 MHz (the later one wins)
        "TDLX": (np.array([65, 130, 300, 310, 800, 2000]) * 1e-9, np.array([-2.0, 0.0, -4.0, -1.5, -6.0, -9.0])),
    }
    payload = {}
    try:
        for name, (d, p) in tables.items():
            cs.ChannelModel.CHANNEL_PROFILES[name] = {"delays": d, "powers": p}
            O.PROFILES[name] = (d, p)
            payload[f"{name}_delays"] = d; payload[f"{name}_powers"] = p
        for name, (T, R, fd, snr, dens, seed) in {"EPA": (2, 2, 70.0, 10.0, 0.1, 611), "TDLX": (2, 1, 150.0, 5.0, 0.05, 612)}.items():
            cfg = {"ofdm": {"fft_size": 1024, "cp_length": 72, "num_symbols": 14, "useful_subcarriers": 600, "subcarrier_spacing": 15000.0},
                   "mimo": {"num_tx_antennas": T, "num_rx_antennas": R}, "channel": {"carrier_freq": 2.0e9}}
            np.random.seed(seed)
            sim = cs.simulate_transmission(cfg, name.lower(), fd, snr, dens)         # (the type is upper-cased, :67)
            pp = sim["pilot_pattern"]
            rx4 = np.repeat(sim["rx_symbols"][:, :, None, :], T, axis=2)
            ls = be.LSEstimator("linear").estimate(rx4, sim["pilot_symbols"], pp.pilot_mask, pp.pilot_positions)
            L = len(tables[name][0])
            np.random.seed(seed)
            dr = O.legacy_frame_draws(np.random, 14, 599, T, R, L, dens)
            o = O.simulate_frame(dict(cfg["ofdm"]), dict(num_tx=T, num_rx=R), name, fd, snr, dr)
            assert np.array_equal(dr["pilot_indices"], pp.pilot_indices)
            for a, b in ((o["channel"], sim["channel"]), (o["rx_symbols"], sim["rx_symbols"]), (o["tx_symbols"], sim["tx_symbols"])):
                assert rel(a, b) <= TOL, name
            cm = cs.ChannelModel(name, fd, 2e9, 15.36e6)
            payload.update({f"{name}_cfg": np.asarray([T, R]), f"{name}_fd": fd, f"{name}_snr": snr, f"{name}_density": dens, f"{name}_seed": seed,
                            f"{name}_delay_samples": cm.delay_samples.astype(np.int64), f"{name}_powers_linear": cm.powers_linear,
                            **{f"{name}_{k}": v for k, v in dr.items()},
                            f"{name}_tx": sim["tx_symbols"][:, 0, :], f"{name}_H": sim["channel"], f"{name}_rx": sim["rx_symbols"],
                            f"{name}_pilot_symbols": sim["pilot_symbols"], f"{name}_H_ls": ls[:, :, 0, :]})
        # the dense public API on the added profile
        np.random.seed(613)
        cm = cs.ChannelModel("TDLX", 150.0, 2e9, 15.36e6)
        h = cm.generate_time_varying_channel(300, 2, 1)
        np.random.seed(613)
        ua = np.empty((6, 2, 1, 20)); up = np.empty_like(ua)
        for l in range(6):
            for t in range(2):
                for r in range(1):
                    ua[l, t, r] = np.random.rand(20); up[l, t, r] = np.random.rand(20)
        assert rel(O.time_varying_channel("TDLX", 150.0, 15.36e6, 300, 2, 1, ua, up), h) <= TOL
        payload.update(TDLX_tv_h=h, TDLX_tv_u_angle=ua, TDLX_tv_u_phase=up)
    finally:
        cs.ChannelModel.CHANNEL_PROFILES.clear(); cs.ChannelModel.CHANNEL_PROFILES.update(saved)
        O.PROFILES.clear(); O.PROFILES.update(o_saved)
    np.savez_compressed(os.path.join(out, "r5_custom_profiles.npz"), numpy_version=np.__version__, **payload)
    print("r5_custom_profiles.npz written")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(args.reference, "src"))
    import baseline_estimators as be
    import channel_simulator as cs
    any_grid(cs, be, args.out)
    custom_profiles(cs, be, args.out)


if __name__ == "__main__":
    main()
