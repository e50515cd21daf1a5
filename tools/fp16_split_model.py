#!/usr/bin/env python3
"""NumPy model of the fp16-split complex GEMM of csrc/mmse_full.hip (cgemm_w_kernel_h2): W h for config 5's frame (P = 1 678, the
golden fixture's pilots and LS estimates, W from the harness covariance) with every REAL operand of the 3-multiplication form split
into two fp16 terms and three fp16 x fp16 products accumulated in float32, against float64 and against the plain float32
3-multiplication form.  CPU only (runs in the build container; needs tests/golden/full_cfg5_4x4_2048_etu.npz).

Measured here: fp16 split 7.3e-7 / 8.0e-7 of max|W h| at 10 / 30 dB, f32 3M 8.7e-7 / 9.8e-7, fp16 hi term alone 1.2e-4."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import cpu_ref as O

g = np.load(os.path.join(ROOT, "tests", "golden", "full_cfg5_4x4_2048_etu.npz"))
pil = g["pilot_indices"]; P = len(pil)
pos, mask = O.pattern_from_indices(pil, 14, 1199)
x = O.unit_symbols(g["u_pilot"])
rx = g["rx"].astype(complex)
h = np.stack([rx[:, r, :][mask] / (x + 1e-12) for r in range(4)]).astype(np.complex64)          # [4, P] LS estimates at the pilots


def split(a, scale):
    a = a.astype(np.float32) * np.float32(scale)
    hi = a.astype(np.float16)
    return hi.astype(np.float32), (a - hi.astype(np.float32)).astype(np.float16).astype(np.float32)


def pow2scale(m):                                   # largest power of two s with 2 m s <= 2^14 (h2_scale of the kernel)
    return 2.0 ** (14 - (np.frexp(2.0 * m)[1]))


for snr in (10.0, 30.0):
    cov = bench.fullcov_covariance(2048, 144, 1200, 4, "ETU", 555.6, pil)
    W = (cov @ np.linalg.inv(cov + 10 ** (-snr / 10) * np.eye(P))).astype(np.complex64)
    ref = (W.astype(complex) @ h.astype(complex).T).T
    sa, sb = pow2scale(np.abs(np.concatenate([W.real, W.imag])).max()), pow2scale(np.abs(np.concatenate([h.real, h.imag])).max())

    def prod(A, B, terms):                          # [i, k] x [m, k]^T, fp16 terms, float32 accumulation
        ah, al = split(A, sa); bh, bl = split(B, sb)
        acc = bh @ ah.T
        if terms == 3:
            acc = (bl @ ah.T) + (bh @ al.T) + acc
        return acc / np.float32(sa * sb)

    def three_m(f):
        p1, p2, p3 = f(W.real, h.real), f(W.imag, h.imag), f((W.real + W.imag).astype(np.float32), (h.real + h.imag).astype(np.float32))
        return (p1 - p2) + 1j * (p3 - p1 - p2)

    sc = np.abs(ref).max()
    e3 = np.abs(three_m(lambda A, B: prod(A, B, 3)) - ref).max() / sc
    e1 = np.abs(three_m(lambda A, B: prod(A, B, 1)) - ref).max() / sc
    e32 = np.abs(three_m(lambda A, B: B.astype(np.float32) @ A.astype(np.float32).T) - ref).max() / sc
    print(f"{snr:.0f} dB: scales W 2^{int(np.log2(sa))} h 2^{int(np.log2(sb))};  max err / max|W h|:  fp16 split (3 terms) {e3:.2e}   f32 3M {e32:.2e}   fp16 hi only {e1:.2e}")
