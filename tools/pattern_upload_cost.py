#!/usr/bin/env python3
"""Host cost of one pattern upload (chest_pattern_set_triangulation: hipMalloc + copy + pack/locate kernels + sync + table
entry) and of the Qhull triangulation that precedes it.  GPU only."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
E = importlib.import_module("channel-estimation-in-5g-network_amd.engine")
IP = importlib.import_module("channel-estimation-in-5g-network_amd.interp_plan")
geom = E.Geometry(); eng = E.Engine(geom, 0); G = geom.grid
pils = [np.sort(np.random.RandomState(i).permutation(G)[:838]) for i in range(300)]
t0 = time.perf_counter()
tris = [IP.linear_triangulation(*np.unravel_index(p, (14, 599))) for p in pils]
t_tri = (time.perf_counter() - t0) / len(pils)
for i in range(20):
    eng.set_pattern(i, pils[i], tris[i])
t0 = time.perf_counter()
for i in range(20, 300):
    eng.set_pattern(i, pils[i], tris[i])
t_up = (time.perf_counter() - t0) / 280
t0 = time.perf_counter()
for i in range(20, 300):
    eng.set_pattern(i, pils[i], tris[i])          # replacing existing slots: adds the sync + free of the old block
t_re = (time.perf_counter() - t0) / 280
print(f"Qhull triangulation {t_tri * 1e3:.2f} ms per pattern;  upload into a new slot {t_up * 1e6:.0f} us;  into an occupied slot {t_re * 1e6:.0f} us")
