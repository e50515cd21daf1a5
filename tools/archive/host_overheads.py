#!/usr/bin/env python3
"""Host-side overheads of the single-frame drop-in path, measured on the GPU box: Qhull Delaunay before / after the GPU
runtime is initialised, and pageable vs pinned host<->device copies of frame-sized arrays."""
import time

import numpy as np
from scipy.spatial import Delaunay


def delaunay_ms(n=30, seed=1):
    S, nsc, P = 14, 599, 838
    rs = np.random.RandomState(seed)
    ts = []
    for _ in range(n):
        idx = np.sort(rs.permutation(S * nsc)[:P])
        ps, pk = np.unravel_index(idx, (S, nsc))
        pts = np.column_stack([ps, pk]).astype(np.float64)
        t0 = time.perf_counter()
        Delaunay(pts).simplices
        ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3, min(ts) * 1e3, max(ts) * 1e3


print("Delaunay(838 pts) ms  med/min/max, before torch import : %.2f %.2f %.2f" % delaunay_ms())
import torch  # noqa: E402
print("                      after torch import             : %.2f %.2f %.2f" % delaunay_ms())
torch.cuda.init()
x = torch.zeros(1 << 20, device="cuda")
torch.cuda.synchronize()
print("                      after GPU init                 : %.2f %.2f %.2f" % delaunay_ms())
torch.set_num_threads(1)
print("                      torch.set_num_threads(1)       : %.2f %.2f %.2f" % delaunay_ms())

for nbytes in (8 * 838, 134176, 268352, 1073408):
    a = np.random.rand(nbytes // 8)
    ta = torch.from_numpy(a)
    pin = torch.empty(nbytes // 8, dtype=torch.float64).pin_memory()
    d = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda")
    def t(f, n=50):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    h2d_page = t(lambda: ta.to("cuda"))
    h2d_pin = t(lambda: (pin.copy_(ta), d.copy_(pin, non_blocking=True)))
    d2h_page = t(lambda: d.cpu())
    d2h_pin = t(lambda: (pin.copy_(d, non_blocking=True), torch.cuda.current_stream().synchronize(), pin.numpy().copy()))
    print(f"{nbytes:8d} B  H2D pageable {h2d_page:7.1f} us  via pinned {h2d_pin:7.1f} us | D2H pageable {d2h_page:7.1f} us  via pinned {d2h_pin:7.1f} us")
