"""CPU oracle for the OFDM channel-simulation + LS/MMSE hot path (float64, NumPy/SciPy).

TEST INFRASTRUCTURE ONLY.  This module is a from-scratch restatement of the arithmetic of the
reference's hot path (``/root/reference/src/channel_simulator.py`` and
``/root/reference/src/baseline_estimators.py``); it exists so that the HIP engine can be checked on
a GPU box where the reference itself is not present.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product package never does.

Parity status: PINNED.  ``tools/make_golden.py`` imports the real reference in the build container,
replays its random draws in reference order, and asserts this restatement reproduces the reference's
outputs to <= 1e-12 before any fixture is written to ``tests/golden/``;
``tests/test_oracle_golden.py`` re-checks the restatement against those committed fixtures.

Every function cites the reference lines it follows (paths relative to ``/root/reference``).
All random inputs are *arguments* (the reference pulls them from the global ``np.random`` stream);
``legacy_frame_draws`` replays the reference's draw order from a ``RandomState``.
"""
from __future__ import annotations

import numpy as np

NUM_OSC = 20  # Jakes oscillators, src/channel_simulator.py:100

# 3GPP TS 36.101 Annex B tap tables as used by src/channel_simulator.py:41-54 (ns, dB).
PROFILES = {
    "EPA": (np.array([0, 30, 70, 90, 110, 190, 410]) * 1e-9,
            np.array([0.0, -1.0, -2.0, -3.0, -8.0, -17.2, -20.8])),
    "EVA": (np.array([0, 30, 150, 310, 370, 710, 1090, 1730, 2510]) * 1e-9,
            np.array([0.0, -1.5, -1.4, -3.6, -0.6, -9.1, -7.0, -12.0, -16.9])),
    "ETU": (np.array([0, 50, 120, 200, 230, 500, 1600, 2300, 5000]) * 1e-9,
            np.array([-1.0, -1.0, -1.0, 0.0, 0.0, 0.0, -3.0, -5.0, -7.0])),
}


# --------------------------------------------------------------------------------------------
# geometry / tables
# --------------------------------------------------------------------------------------------
def tap_table(model_type: str, sampling_rate: float):
    """(powers_linear[L], delay_samples[L]) -- src/channel_simulator.py:67-82.

    Power is normalised to sum 1 (:75-78); delays are rounded half-to-even to samples (:81).
    Unknown type raises KeyError (:72); the type is upper-cased (:67).
    """
    delays, p_db = PROFILES[model_type.upper()]
    p = 10 ** (p_db / 10)
    p = p / np.sum(p)
    d = np.round(delays * sampling_rate).astype(int)
    return p, d


def used_indices(fft_size: int, useful_subcarriers: int) -> np.ndarray:
    """fftshift-ed bin indices of the used subcarriers, DC removed -- src/channel_simulator.py:138-148."""
    dc = fft_size // 2
    idx = np.arange(dc - useful_subcarriers // 2, dc + useful_subcarriers // 2)
    return idx[idx != dc]


def pattern_from_indices(pilot_indices: np.ndarray, num_symbols: int, num_sc: int):
    """(positions, mask) for sorted flat pilot indices -- src/channel_simulator.py:231-236."""
    pos = np.unravel_index(pilot_indices, (num_symbols, num_sc))
    mask = np.zeros((num_symbols, num_sc), dtype=bool)
    mask[pos] = True
    return pos, mask


def num_pilots(num_symbols: int, num_sc: int, density: float) -> int:
    """int() truncation of the pilot count -- src/channel_simulator.py:223-224."""
    return int(num_sc * num_symbols * density)


# --------------------------------------------------------------------------------------------
# random-draw replay (legacy MT19937 stream, reference order)
# --------------------------------------------------------------------------------------------
def legacy_frame_draws(rs, num_symbols, num_sc, num_tx, num_rx, num_paths, density):
    """Consume ``rs`` (np.random.RandomState or the np.random module) exactly as one
    ``simulate_transmission`` call does -- src/channel_simulator.py:227-229 (shuffle), :395 (pilot
    phases), :399 (data phases), :102-110 (per path->tx->rx: 20 angles then 20 phases),
    :342 (noise real block, then imaginary block).  Returns the raw draws (uniforms in [0,1) are
    stored *before* the 2*pi scaling so that they can be injected into any implementation).
    """
    total = num_symbols * num_sc
    P = num_pilots(num_symbols, num_sc, density)
    perm = np.arange(total)
    rs.shuffle(perm)
    pilot_indices = np.sort(perm[:P])
    two_pi = 2 * np.pi
    # uniform(0, 2pi, n) == 0 + 2pi*random_sample(n): recover the underlying uniforms exactly.
    u_pilot = rs.random_sample(P)
    u_data = rs.random_sample(total - P)
    u_angle = np.empty((num_paths, num_tx, num_rx, NUM_OSC))
    u_phase = np.empty((num_paths, num_tx, num_rx, NUM_OSC))
    for l in range(num_paths):
        for t in range(num_tx):
            for r in range(num_rx):
                u_angle[l, t, r] = rs.random_sample(NUM_OSC)
                u_phase[l, t, r] = rs.random_sample(NUM_OSC)
    n_re = rs.standard_normal((num_symbols, num_rx, num_sc))
    n_im = rs.standard_normal((num_symbols, num_rx, num_sc))
    del two_pi
    return dict(pilot_indices=pilot_indices, u_pilot=u_pilot, u_data=u_data,
                u_angle=u_angle, u_phase=u_phase, n_re=n_re, n_im=n_im)


# --------------------------------------------------------------------------------------------
# a3 / a9: Jakes fading and frequency response
# --------------------------------------------------------------------------------------------
def jakes_gains(u_angle, u_phase, doppler_hz, sampling_rate, sample_idx):
    """Complex fading gain g[..., n_time] for every (path, tx, rx) at the given sample indices.

    src/channel_simulator.py:109-124: angles = 2pi*u, phases = 2pi*u', t = idx/fs,
    g = sum_n exp(j(2pi*fd*cos(angle_n)*t + phase_n)) / sqrt(2*20).
    """
    angles = 2 * np.pi * np.asarray(u_angle)
    phases = 2 * np.pi * np.asarray(u_phase)
    t = np.asarray(sample_idx) / sampling_rate
    shift = doppler_hz * np.cos(angles)                      # (..., 20)
    arg = (2 * np.pi * shift)[..., None] * t + phases[..., None]   # (..., 20, n_time)
    g = np.cos(arg).sum(axis=-2) + 1j * np.sin(arg).sum(axis=-2)
    return g / np.sqrt(2 * NUM_OSC)


def impulse_response(gains, powers, delay_samples):
    """h[time, rx, tx, D+1] from g[path, tx, rx, time].

    src/channel_simulator.py:96-125.  Plain assignment per path, in path order: when two paths round
    to the same delay the later one overwrites the earlier (quirk Q1).
    """
    L, T, R, nt = gains.shape
    D = int(np.max(delay_samples))
    h = np.zeros((nt, R, T, D + 1), dtype=complex)
    for l in range(L):
        amp = np.sqrt(powers[l])
        h[:, :, :, delay_samples[l]] = amp * np.transpose(gains[l], (2, 1, 0))
    return h


def time_varying_channel(model_type, doppler_hz, sampling_rate, num_samples, num_tx, num_rx,
                         u_angle, u_phase):
    """Dense public API ChannelModel.generate_time_varying_channel -- src/channel_simulator.py:84-127."""
    p, d = tap_table(model_type, sampling_rate)
    g = jakes_gains(u_angle, u_phase, doppler_hz, sampling_rate, np.arange(num_samples))
    return impulse_response(g, p, d)


def frequency_response(h_sym, fft_size, used):
    """H[s, r, t, k] = fftshift(fft(h[s, r, t, :], n=N))[used] -- src/channel_simulator.py:300-309."""
    Hf = np.fft.fftshift(np.fft.fft(h_sym, n=fft_size, axis=-1), axes=-1)
    return Hf[..., used]


def channel_frequency_response(model_type, doppler_hz, fft_size, cp_length, num_symbols,
                               useful_subcarriers, subcarrier_spacing, u_angle, u_phase):
    """MIMOChannel.generate_channel_frequency_response -- src/channel_simulator.py:274-311.

    The impulse response is sampled at each symbol's FIRST sample only (:301-302, quirk Q2).
    """
    fs = fft_size * subcarrier_spacing
    p, d = tap_table(model_type, fs)
    sample_idx = np.arange(num_symbols) * (fft_size + cp_length)
    g = jakes_gains(u_angle, u_phase, doppler_hz, fs, sample_idx)
    h = impulse_response(g, p, d)
    return frequency_response(h, fft_size, used_indices(fft_size, useful_subcarriers))


# --------------------------------------------------------------------------------------------
# a5 / a6: OFDM modulation
# --------------------------------------------------------------------------------------------
def ofdm_modulate(symbols, fft_size, cp_length, useful_subcarriers):
    """(S, nsc) -> (S, N+cp): scatter, ifft(ifftshift)*sqrt(N), prepend CP -- src/channel_simulator.py:160-178."""
    used = used_indices(fft_size, useful_subcarriers)
    S = symbols.shape[0]
    grid = np.zeros((S, fft_size), dtype=complex)
    grid[:, used] = symbols
    td = np.fft.ifft(np.fft.ifftshift(grid, axes=-1), axis=-1) * np.sqrt(fft_size)
    return np.concatenate([td[:, fft_size - cp_length:], td], axis=1)


def ofdm_demodulate(rx, fft_size, cp_length, useful_subcarriers):
    """(S, N+cp) -> (S, nsc): drop CP, fftshift(fft)/sqrt(N), gather -- src/channel_simulator.py:190-203."""
    used = used_indices(fft_size, useful_subcarriers)
    fd = np.fft.fftshift(np.fft.fft(rx[:, cp_length:], axis=-1), axes=-1) / np.sqrt(fft_size)
    return fd[:, used]


# --------------------------------------------------------------------------------------------
# a8 / a10 / a11: grid, channel application, one frame
# --------------------------------------------------------------------------------------------
def unit_symbols(u):
    """exp(j*uniform(0,2pi)) from the underlying uniforms -- src/channel_simulator.py:395,399."""
    return np.exp(1j * (0.0 + (2 * np.pi - 0.0) * np.asarray(u)))


def insert_pilots(mask, data_symbols, pilot_symbols):
    """Row-major boolean-mask scatter -- src/channel_simulator.py:249-252."""
    grid = np.zeros(mask.shape, dtype=complex)
    grid[mask] = pilot_symbols
    grid[~mask] = data_symbols
    return grid


def apply_channel(tx, H, snr_db, n_re, n_im):
    """y = Hx per RE; noise power from the realised frame power; real block then imaginary block
    -- src/channel_simulator.py:326-345."""
    y = np.einsum("srtk,stk->srk", H, tx)
    signal_power = np.mean(np.abs(y) ** 2)
    noise_power = signal_power / (10 ** (snr_db / 10))
    noise_std = np.sqrt(noise_power / 2)
    return y + (n_re + 1j * n_im) * noise_std


def simulate_frame(ofdm, mimo, channel_type, doppler_hz, snr_db, draws):
    """One ``simulate_transmission`` call with injected draws -- src/channel_simulator.py:348-421.

    ``ofdm`` = dict(fft_size, cp_length, num_symbols, useful_subcarriers, subcarrier_spacing);
    ``mimo`` = dict(num_tx, num_rx).  Every TX antenna sends the same grid (:402-404, quirk Q5).
    """
    N, cp, S = ofdm["fft_size"], ofdm["cp_length"], ofdm["num_symbols"]
    U, scs = ofdm["useful_subcarriers"], ofdm["subcarrier_spacing"]
    T, R = mimo["num_tx"], mimo["num_rx"]
    nsc = len(used_indices(N, U))
    pos, mask = pattern_from_indices(draws["pilot_indices"], S, nsc)
    pilots = unit_symbols(draws["u_pilot"])
    data = unit_symbols(draws["u_data"])
    grid = insert_pilots(mask, data, pilots)
    tx = np.repeat(grid[:, None, :], T, axis=1)
    H = channel_frequency_response(channel_type, doppler_hz, N, cp, S, U, scs,
                                   draws["u_angle"], draws["u_phase"])
    rx = apply_channel(tx, H, snr_db, draws["n_re"], draws["n_im"])
    return dict(tx_symbols=tx, rx_symbols=rx, channel=H, pilot_symbols=pilots,
                pilot_indices=draws["pilot_indices"], pilot_positions=pos, pilot_mask=mask)


# --------------------------------------------------------------------------------------------
# a12-a16: estimators
# --------------------------------------------------------------------------------------------
def _griddata_complex(h_pilots, positions, grid_shape, method):
    """Real and imaginary parts interpolated separately with scipy griddata, unscaled integer
    coordinates, fill 0 outside the hull -- src/baseline_estimators.py:57-81 (same at :206-230)."""
    from scipy.interpolate import griddata
    S, nsc = grid_shape
    pts = np.column_stack([positions[0], positions[1]])
    ss, kk = np.meshgrid(np.arange(S), np.arange(nsc), indexing="ij")
    xi = np.column_stack([ss.ravel(), kk.ravel()])        # row-major (sym outer, sc inner)
    re = griddata(pts, h_pilots.real, xi, method=method, fill_value=0.0)
    im = griddata(pts, h_pilots.imag, xi, method=method, fill_value=0.0)
    return (re + 1j * im).reshape(S, nsc)


def ls_estimate(rx_4d, tx_pilots, mask, positions, method="linear"):
    """LSEstimator.estimate -- src/baseline_estimators.py:97-117: per (rx, tx) pilot division
    y/(x+1e-12) (:110) then scattered interpolation."""
    S, R, T, nsc = rx_4d.shape
    out = np.zeros(rx_4d.shape, dtype=complex)
    for r in range(R):
        for t in range(T):
            h_p = rx_4d[:, r, t, :][mask] / (tx_pilots + 1e-12)
            out[:, r, t, :] = _griddata_complex(h_p, positions, (S, nsc), method)
    return out


def mmse_at_pilots(h_ls, snr_db, covariance=None, estimate_statistics=True, dense=True):
    """MMSEEstimator.estimate_at_pilots -- src/baseline_estimators.py:173-196.

    sigma^2 = 1/10^(snr/10) (:174-175); default R = mean|h_ls|^2 * I (:177-180);
    h = R (R + sigma^2 I)^-1 h_ls (:185-190).  ``dense=False`` uses the algebraically identical
    scalar shrinkage for the diagonal case.
    """
    sigma2 = 1.0 / (10 ** (snr_db / 10))
    n = len(h_ls)
    if estimate_statistics or covariance is None:
        p = np.mean(np.abs(h_ls) ** 2)
        if not dense:
            return (p / (p + sigma2)) * h_ls
        Rh = np.eye(n) * p
    else:
        Rh = covariance
    Ry = Rh + sigma2 * np.eye(n)
    return (Rh @ np.linalg.inv(Ry)) @ h_ls


def mmse_estimate(rx_4d, tx_pilots, mask, positions, snr_db, covariance=None,
                  estimate_statistics=True, dense=True):
    """MMSEEstimator.estimate -- src/baseline_estimators.py:248-270; interpolation is always
    'linear' (:266-268, quirk Q10)."""
    S, R, T, nsc = rx_4d.shape
    out = np.zeros(rx_4d.shape, dtype=complex)
    for r in range(R):
        for t in range(T):
            h_ls = rx_4d[:, r, t, :][mask].reshape(-1) / (tx_pilots + 1e-12)
            h_p = mmse_at_pilots(h_ls, snr_db, covariance, estimate_statistics, dense)
            out[:, r, t, :] = _griddata_complex(h_p, positions, (S, nsc), "linear")
    return out


def repeat_over_tx(rx, num_tx):
    """rx (S,R,nsc) -> rx_4d (S,R,T,nsc) as every caller builds it -- src/dataset_generator.py:63-64."""
    return np.repeat(rx[:, :, None, :], num_tx, axis=2)


def equalize(rx, H_est, method="zf"):
    """equalize_channel -- src/baseline_estimators.py:286-312 (ZF with 1e-8 loading, MMSE with
    fixed noise_var 0.01; unknown method -> ValueError)."""
    if method == "zf":
        load = 1e-8
    elif method == "mmse":
        load = 0.01
    else:
        raise ValueError(f"Unknown equalization method: {method}")
    S, R, T, nsc = H_est.shape
    Hm = np.transpose(H_est, (0, 3, 1, 2))                       # (S, nsc, R, T)
    Hh = np.conj(np.swapaxes(Hm, -1, -2))
    G = Hh @ Hm + load * np.eye(T)
    W = np.linalg.inv(G) @ Hh
    y = np.transpose(rx, (0, 2, 1))[..., None]                   # (S, nsc, R, 1)
    return np.transpose((W @ y)[..., 0], (0, 2, 1))


def evaluate_estimator(H_true, H_est):
    """mse / nmse / nmse_db with the +1e-12 guards -- src/baseline_estimators.py:326-337."""
    mse = np.mean(np.abs(H_true - H_est) ** 2)
    nmse = mse / (np.mean(np.abs(H_true) ** 2) + 1e-12)
    return {"mse": mse, "nmse": nmse, "nmse_db": 10 * np.log10(nmse + 1e-12)}


# --------------------------------------------------------------------------------------------
# QPSK / 16-QAM / BER helpers -- src/utils.py:71-158
# --------------------------------------------------------------------------------------------
def qam_tables(M):
    """(constellation, gray_map) exactly as src/utils.py:92-103 and :125-136 build them; other orders raise
    NotImplementedError (:104, :138)."""
    if M == 4:
        return np.array([1+1j, -1+1j, 1-1j, -1-1j]) / np.sqrt(2), np.array([0, 1, 3, 2])
    if M == 16:
        pts = np.array([-3-3j, -3-1j, -3+3j, -3+1j, -1-3j, -1-1j, -1+3j, -1+1j,
                        3-3j, 3-1j, 3+3j, 3+1j, 1-3j, 1-1j, 1+3j, 1+1j]) / np.sqrt(10)
        return pts, np.array([0, 1, 3, 2, 4, 5, 7, 6, 12, 13, 15, 14, 8, 9, 11, 10])
    raise NotImplementedError(f"Modulation order {M} not implemented")


def qam_modulate(bits, M=4):
    """qam_modulation -- src/utils.py:71-108.  The reference indexes a Python *list* gray_map with an array (:106),
    which raises TypeError under every NumPy release this project can run on, so the function itself cannot be
    executed; this restates the evident intent (gray_map as an array).  The mapping is pinned through the inverse
    the reference CAN execute: qam_demodulation(constellation points) in tests/golden/qam_kat.json."""
    pts, gray = qam_tables(M)
    bps = int(np.log2(M))
    bits = np.asarray(bits)
    n = len(bits) // bps
    dec = bits[:n * bps].reshape(-1, bps) @ (2 ** np.arange(bps)[::-1])
    return pts[gray[dec]]


def qam_demodulate(symbols, M=4):
    """qam_demodulation -- src/utils.py:111-153: nearest point by |.|, first minimum (np.argmin), inverse Gray map
    (np.argsort(gray_map)), bits MSB first, flattened."""
    pts, gray = qam_tables(M)
    bps = int(np.log2(M))
    symbols = np.asarray(symbols)
    det = np.argmin(np.abs(symbols[:, None] - pts[None, :]), axis=1)
    dec = np.argsort(gray)[det]
    return ((dec[:, None] >> np.arange(bps)[::-1]) & 1).astype(int).reshape(-1)


def bit_error_rate(tx_bits, rx_bits):
    """calculate_ber -- src/utils.py:156-158."""
    tx_bits, rx_bits = np.asarray(tx_bits), np.asarray(rx_bits)
    return np.sum(tx_bits != rx_bits) / len(tx_bits)


# --------------------------------------------------------------------------------------------
# "faithful" mode pieces: same algorithmic structure (and cost profile) as the reference, used
# only for the cpu_baseline timing in bench.py.
# ---- training-side feature packing --------------------------------------------------------------------------
def ml_first_pair(rx_symbols, H_ls, H_true):
    """The slices every reference loader works on: rx[:, :, 0, :], H_ls[:, :, 0, 0, :], H_true[:, :, 0, 0, :]
    (src/train.py:43-46, :66-68; run_phase4_training.py:68-69, :84-86)."""
    return np.asarray(rx_symbols)[:, :, 0, :], np.asarray(H_ls)[:, :, 0, 0, :], np.asarray(H_true)[:, :, 0, 0, :]


def ml_train_stats(rx_symbols, H_ls, H_true):
    """(rx_mean, rx_std, H_ls_mean, H_ls_std, H_true_mean, H_true_std) of train.ChannelDataset._compute_normalization_stats
    (src/train.py:41-58): the mean of the real-part and the imaginary-part statistic, population std; float64 here."""
    out = []
    for x in ml_first_pair(rx_symbols, H_ls, H_true):
        re, im = x.real.astype(np.float64), x.imag.astype(np.float64)
        out += [0.5 * (re.mean() + im.mean()), 0.5 * (re.std() + im.std())]
    return np.array(out)


def ml_phase4_stats(rx_symbols, H_ls, H_true):
    """(rx_std, H_ls_std, H_true_std) of run_phase4_training.ChannelEstimationDataset._compute_stats (:65-76): std(|x|) + 1e-8."""
    return np.array([np.std(np.abs(x.astype(np.complex128))) + 1e-8 for x in ml_first_pair(rx_symbols, H_ls, H_true)])


def ml_pack_training(rx_symbols, H_ls, H_true, pilot_mask, mode="train"):
    """inputs [N,5,S,nsc], targets [N,2,S,nsc] (float64) as the loaders' __getitem__ stack them (src/train.py:63-94;
    run_phase4_training.py:81-112): channels (rx re, rx im, H_ls re, H_ls im, mask); mode "raw" / "train" / "phase4"."""
    rx, ls, ht = (x.astype(np.complex128) for x in ml_first_pair(rx_symbols, H_ls, H_true))
    ri = lambda z: np.stack([z.real, z.imag], axis=1)
    a, b, c = ri(rx), ri(ls), ri(ht)
    if mode == "train":
        st = ml_train_stats(rx_symbols, H_ls, H_true)
        a, b, c = (a - st[0]) / (st[1] + 1e-8), (b - st[2]) / (st[3] + 1e-8), (c - st[4]) / (st[5] + 1e-8)
    elif mode == "phase4":
        st = ml_phase4_stats(rx_symbols, H_ls, H_true)
        a, b, c = a / st[0], b / st[1], c / st[2]
    elif mode != "raw":
        raise ValueError(mode)
    return np.concatenate([a, b, np.asarray(pilot_mask, dtype=np.float64)[:, None]], axis=1), c


# --------------------------------------------------------------------------------------------
def channel_frequency_response_faithful(model_type, doppler_hz, fft_size, cp_length, num_symbols,
                                        useful_subcarriers, subcarrier_spacing, u_angle, u_phase):
    """As channel_frequency_response but synthesising ALL num_symbols*(N+cp) time samples per tap,
    one oscillator at a time, before picking the symbol-start samples (the reference's cost
    structure, src/channel_simulator.py:112-125 and :285-309)."""
    fs = fft_size * subcarrier_spacing
    p, d = tap_table(model_type, fs)
    L, T, R, _ = u_angle.shape
    sps = fft_size + cp_length
    ns = num_symbols * sps
    D = int(d.max())
    h = np.zeros((ns, R, T, D + 1), dtype=complex)
    tvec = np.arange(ns) / fs
    for l in range(L):
        for t in range(T):
            for r in range(R):
                ang = 2 * np.pi * u_angle[l, t, r]
                ph = 2 * np.pi * u_phase[l, t, r]
                acc_re = np.zeros(ns)
                acc_im = np.zeros(ns)
                for n in range(NUM_OSC):
                    w = 2 * np.pi * (doppler_hz * np.cos(ang[n]))
                    acc_re += np.cos(w * tvec + ph[n])
                    acc_im += np.sin(w * tvec + ph[n])
                h[:, r, t, d[l]] = np.sqrt(p[l]) * (acc_re + 1j * acc_im) / np.sqrt(2 * NUM_OSC)
    used = used_indices(fft_size, useful_subcarriers)
    H = np.zeros((num_symbols, R, T, len(used)), dtype=complex)
    for s in range(num_symbols):
        for r in range(R):
            for t in range(T):
                H[s, r, t] = np.fft.fftshift(np.fft.fft(h[s * sps, r, t], n=fft_size))[used]
    return H


def apply_channel_faithful(tx, H, snr_db, n_re, n_im):
    """Per-RE matrix-vector loop, the reference's structure (src/channel_simulator.py:330-334)."""
    S, R, T, nsc = H.shape
    y = np.zeros((S, R, nsc), dtype=complex)
    for s in range(S):
        for k in range(nsc):
            y[s, :, k] = H[s, :, :, k] @ tx[s, :, k]
    noise_std = np.sqrt(np.mean(np.abs(y) ** 2) / (10 ** (snr_db / 10)) / 2)
    return y + (n_re + 1j * n_im) * noise_std
