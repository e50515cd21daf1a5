"""CPU restatement of the engine's counter-based random streams (TEST INFRASTRUCTURE ONLY).

Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11) keyed exactly as
``csrc/chest_device.h`` does: counter = (index, stream, frame_lo, frame_hi), key = (seed_lo, seed_hi).
``native_frame_draws`` returns the same dict of draws as ``cpu_ref.legacy_frame_draws`` so the float64
oracle can be run on the engine's own random inputs.  This has no counterpart in the reference (which
uses the sequential global MT19937 stream); it pins the native-RNG mode of the engine to the oracle.
"""
from __future__ import annotations

import numpy as np

STREAM_OSC, STREAM_SYM, STREAM_NOISE = 1, 2, 3
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over ``c0`` (uint32 array); other counter words / key words are scalars."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.full_like(c0, c1)
    c2 = np.full_like(c0, c2)
    c3 = np.full_like(c0, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def draw(seed, frame, stream, index):
    seed = int(seed) & (2**64 - 1)
    frame = int(frame) & (2**64 - 1)
    return philox4x32_10(np.asarray(index, dtype=np.uint64), stream, frame & 0xFFFFFFFF, frame >> 32,
                         seed & 0xFFFFFFFF, seed >> 32)


def u01(x):
    """24-bit uniform in [0,1), as float32 exactly like the device, returned as float64."""
    return (x >> np.uint32(8)).astype(np.float64) * 2.0 ** -24


def u01_open0(x):
    return ((x >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24


def gauss_pair(a, b):
    mag = np.sqrt(-2.0 * np.log(u01_open0(a)))
    ang = 2.0 * np.pi * u01(b)
    return mag * np.cos(ang), mag * np.sin(ang)


def native_frame_draws(seed, frame, pilot_indices, num_symbols, num_sc, num_tx, num_rx, num_paths):
    """The engine's draws for global frame index ``frame`` in the oracle's input format."""
    S, nsc, T, R, L = num_symbols, num_sc, num_tx, num_rx, num_paths
    G = S * nsc
    # symbol phase of RE (s, k): word (k & 1) of draw(index = s*npair + (k >> 1)), npair = ceil(nsc/2)
    npair = (nsc + 1) // 2
    sw = draw(seed, frame, STREAM_SYM, np.arange(S * npair))
    u_re = np.stack([u01(sw[0]), u01(sw[1])], axis=-1).reshape(S, npair * 2)[:, :nsc]
    mask = np.zeros(G, dtype=bool)
    mask[np.asarray(pilot_indices)] = True
    u_flat = u_re.reshape(-1)
    n_osc = L * T * R * 20
    ox, oy, _, _ = draw(seed, frame, STREAM_OSC, np.arange(n_osc))
    u_angle = u01(ox).reshape(L, T, R, 20)
    u_phase = u01(oy).reshape(L, T, R, 20)
    # noise of RE (s, r, k): words (0,1) [k even] or (2,3) [k odd] of draw(index = (s*R + r)*npair + (k >> 1))
    w = draw(seed, frame, STREAM_NOISE, np.arange(S * R * npair))
    re0, im0 = gauss_pair(w[0], w[1])
    re1, im1 = gauss_pair(w[2], w[3])
    n_re = np.stack([re0, re1], axis=-1).reshape(S, R, npair * 2)[:, :, :nsc]
    n_im = np.stack([im0, im1], axis=-1).reshape(S, R, npair * 2)[:, :, :nsc]
    return dict(pilot_indices=np.asarray(pilot_indices), u_pilot=u_flat[mask], u_data=u_flat[~mask], u_angle=u_angle,
                u_phase=u_phase, n_re=n_re, n_im=n_im)


STREAM_PATTERN = 4


def native_pilot_pattern(seed, pattern_id, grid, num_pilots):
    """Pilot set of a device-drawn pattern (csrc/pattern_kernels.hip step 1): resource element g gets the key
    word (g & 3) of draw(seed, frame = pattern_id, STREAM_PATTERN, index = g >> 2); the num_pilots smallest keys, ties by
    element index, are the pilots.  Returns the sorted flat indices -- a uniformly random subset, the distribution of
    PilotPattern's shuffle / take / sort (src/channel_simulator.py:224-229)."""
    w = draw(seed, pattern_id, STREAM_PATTERN, np.arange((grid + 3) // 4))
    keys = np.stack(w, axis=1).reshape(-1)[:grid].astype(np.uint64)
    order = np.lexsort((np.arange(grid), keys))            # by key, ties by element index
    return np.sort(order[:num_pilots]).astype(np.int32)
