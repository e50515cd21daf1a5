"""The reference's own executable test scripts (test_phase1_channels.py, test_phase1_transmission.py,
test_phase2_ls.py, test_phase2_mmse.py) re-expressed as pytest cases: the same calls through the same
import paths (``from src.channel_simulator import ...``), the same assertions (file:line cited).  Like the originals
they are unseeded apart from a fixed seed for repeatability; value-level parity lives in test_gpu_dropin.py."""
import numpy as np
import pytest

from src.baseline_estimators import LSEstimator, MMSEEstimator, evaluate_estimator
from src.channel_simulator import ChannelModel, MIMOConfig, OFDMConfig, simulate_transmission

pytestmark = pytest.mark.gpu

CONFIG = {   # the keys of configs/experiment_config.yaml the scripts read
    "ofdm": {"fft_size": 1024, "cp_length": 72, "subcarrier_spacing": 15000, "num_symbols": 14, "useful_subcarriers": 600},
    "mimo": {"num_tx_antennas": 2, "num_rx_antennas": 2},
    "channel": {"models": ["EPA", "EVA", "ETU"], "doppler_hz": [10, 50, 100, 200], "carrier_freq": 2.0e9},
}


def _nmse_db(h_true, h_est):                      # utils.calculate_nmse + linear2db (src/utils.py:43-46, 165-170)
    nmse = np.mean(np.abs(h_true - h_est) ** 2) / np.mean(np.abs(h_true) ** 2)
    return 10 * np.log10(nmse + 1e-12)


@pytest.mark.parametrize("model,doppler", [("EPA", 10), ("EVA", 50), ("ETU", 200)])
def test_phase1_channels(model, doppler):
    """test_phase1_channels.py:33-88."""
    np.random.seed(1)
    ofdm, mimo = OFDMConfig(), MIMOConfig()
    fs = ofdm.fft_size * ofdm.subcarrier_spacing
    cm = ChannelModel(model, doppler, 2.0e9, fs)
    num_samples = ofdm.num_symbols * ofdm.fft_size                   # 14 336, as the script uses
    H = cm.generate_time_varying_channel(num_samples, mimo.num_tx, mimo.num_rx)
    max_delay = int(np.max(cm.delay_samples))
    assert H.shape == (num_samples, mimo.num_rx, mimo.num_tx, max_delay + 1)          # :67-68
    assert not np.isnan(H).any() and not np.isinf(H).any()                             # :83-84
    assert np.max(np.abs(H)) > 0                                                        # :88
    assert np.iscomplexobj(H)
    H2 = cm.generate_time_varying_channel(500, 1, 1)                                    # :148, :166-170
    assert H2.shape == (500, 1, 1, max_delay + 1)


@pytest.mark.parametrize("channel,doppler,snr", [("EPA", 10, 20), ("EVA", 50, 10), ("ETU", 200, 0)])
def test_phase1_transmission(channel, doppler, snr):
    """test_phase1_transmission.py:50-106."""
    np.random.seed(2)
    pilot_density = 0.10
    out = simulate_transmission(config=CONFIG, channel_type=channel, snr_db=snr, doppler_hz=doppler, pilot_density=pilot_density)
    S, T, R, nsc = 14, 2, 2, 600 - 1                                                    # DC removed (:71)
    assert out["rx_symbols"].shape == (S, R, nsc) and out["tx_symbols"].shape == (S, T, nsc)   # :85-86
    assert out["channel"].shape == (S, R, T, nsc) and out["pilot_pattern"].pilot_mask.shape == (S, nsc)   # :87-88
    mask = out["pilot_pattern"].pilot_mask
    assert abs(mask.sum() / mask.size - pilot_density) < 0.05                          # :101
    assert not np.isnan(out["rx_symbols"]).any() and not np.isinf(out["rx_symbols"]).any()   # :105-106
    assert out["snr_db"] == snr


@pytest.mark.parametrize("T,R", [(1, 1), (2, 2), (4, 4)])
def test_phase1_antenna_configurations(T, R):
    """test_phase1_transmission.py:181-209."""
    np.random.seed(3)
    cfg = dict(CONFIG, mimo={"num_tx_antennas": T, "num_rx_antennas": R})
    out = simulate_transmission(cfg, "EVA", 50, 15, 0.1)
    assert out["rx_symbols"].shape[1] == R and out["tx_symbols"].shape[1] == T
    assert out["channel"].shape[1:3] == (R, T)


@pytest.mark.parametrize("snr,density,interp", [(20, 0.10, "linear"), (15, 0.10, "cubic"), (10, 0.05, "linear")])
def test_phase2_ls(snr, density, interp):
    """test_phase2_ls.py:27-106 (including its 'cubic' case)."""
    np.random.seed(4)
    out = simulate_transmission(config=CONFIG, channel_type="EVA", snr_db=snr, doppler_hz=50, pilot_density=density)
    rx, tx, H_true, pp = out["rx_symbols"], out["tx_symbols"], out["channel"], out["pilot_pattern"]
    mask = pp.pilot_mask
    positions = np.where(mask)                                                          # :62
    S, R, nsc = rx.shape
    T = tx.shape[1]
    rx4 = np.zeros((S, R, T, nsc), dtype=complex)                                       # :66-73
    for t in range(T):
        rx4[:, :, t, :] = rx
    H_full = LSEstimator(interpolation_method=interp).estimate(rx_symbols=rx4, tx_pilots=out["pilot_symbols"],
                                                               pilot_mask=mask, pilot_positions=positions)
    H_est, H_slice = H_full[:, 0, 0, :], H_true[:, 0, 0, :]
    assert H_est.shape == H_slice.shape                                                 # :100
    assert not np.isnan(H_est).any() and not np.isinf(H_est).any()                      # :101-102
    assert _nmse_db(H_slice, H_est) < 10                                                # the script only warns (:105-106)


@pytest.mark.parametrize("snr,density", [(20, 0.10), (10, 0.10), (5, 0.05)])
def test_phase2_mmse(snr, density):
    """test_phase2_mmse.py: same flow with MMSEEstimator, asserts at :98-100; MMSE should not be worse than LS."""
    np.random.seed(5)
    out = simulate_transmission(config=CONFIG, channel_type="EVA", snr_db=snr, doppler_hz=50, pilot_density=density)
    pp = out["pilot_pattern"]
    rx4 = np.repeat(out["rx_symbols"][:, :, None, :], 2, axis=2)
    est = MMSEEstimator(estimate_statistics=True)
    H_full = est.estimate(rx_symbols=rx4, tx_pilots=out["pilot_symbols"], pilot_mask=pp.pilot_mask,
                          pilot_positions=pp.pilot_positions, snr_db=snr)
    H_est, H_slice = H_full[:, 0, 0, :], out["channel"][:, 0, 0, :]
    assert H_est.shape == H_slice.shape and not np.isnan(H_est).any() and not np.isinf(H_est).any()
    ls = LSEstimator("linear").estimate(rx4, out["pilot_symbols"], pp.pilot_mask, pp.pilot_positions)
    m_ls, m_mm = evaluate_estimator(out["channel"], ls), evaluate_estimator(out["channel"], H_full)
    assert m_mm["mse"] <= m_ls["mse"] * 1.0001                                          # shrinkage never hurts against H_true here
