#!/usr/bin/env python3
"""Does Qhull's choice inside co-circular lattice quadrilaterals follow a LOCAL rule the device triangulator could copy?  (CPU only.)
For random 10 %-density pilot patterns on 14 x 599: every interior Delaunay edge of scipy.spatial.Delaunay whose exact in-circle test is 0
(the two triangles' four pilots are co-circular: either diagonal is Delaunay), and how often candidate rules name the diagonal Qhull
chose.  Round 5: 853 such edges in 12 patterns; largest / smallest point index 0.48 / 0.49, largest / smallest |p|^2 0.53 / 0.50, positive
slope 0.42, the shorter diagonal 0.22 (equal lengths 0.47): no better than a coin.  Qhull triangulates a merged co-circular facet as a
fan from its highest-numbered vertex, and vertex numbers follow the order in which its beneath-beyond pass happened to add the points.
    python tools/qhull_tie_rules.py [seed=0] [patterns=12]"""
import numpy as np, sys
from scipy.spatial import Delaunay
S, NSC = 14, 599
def incircle(a, b, c, d):
    # >0 if d inside circumcircle of ccw (a,b,c); exact in Python ints
    ax, ay = a[0]-d[0], a[1]-d[1]; bx, by = b[0]-d[0], b[1]-d[1]; cx, cy = c[0]-d[0], c[1]-d[1]
    return (ax*ax+ay*ay)*(bx*cy-cx*by) - (bx*bx+by*by)*(ax*cy-cx*ay) + (cx*cx+cy*cy)*(ax*by-bx*ay)
def orient(a,b,c): return (b[0]-a[0])*(c[1]-a[1]) - (b[1]-a[1])*(c[0]-a[0])
rows = []
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for pat in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    G = S * NSC
    idx = np.sort(rs.permutation(G)[: G // 10])
    pts = np.stack([idx // NSC, idx % NSC], 1).astype(np.int64)      # (symbol, subcarrier): griddata's point order
    tri = Delaunay(pts.astype(float))
    simp, nbr = tri.simplices, tri.neighbors
    for t in range(len(simp)):
        for i in range(3):
            u = nbr[t, i]
            if u < 0 or u < t: continue
            b, c = simp[t, (i+1) % 3], simp[t, (i+2) % 3]; a = simp[t, i]
            d = [v for v in simp[u] if v != b and v != c][0]
            A, B, C, D = (tuple(int(x) for x in pts[v]) for v in (a, b, c, d))
            if orient(A, B, C) < 0: B, C, b, c = C, B, c, b
            if incircle(A, B, C, D) == 0:
                rows.append((pat, a, b, c, d) + A + B + C + D)       # chosen diagonal: (b, c); alternative: (a, d)
rows = np.asarray(rows, dtype=np.int64)
print(len(rows), "co-circular interior edges")
pat, a, b, c, d = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 4]
A, B, C, D = rows[:, 5:7], rows[:, 7:9], rows[:, 9:11], rows[:, 11:13]
h = lambda P: P[:, 0] ** 2 + P[:, 1] ** 2
dx, dy = C[:, 0] - B[:, 0], C[:, 1] - B[:, 1]
lb = dx * dx + dy * dy
la = (D[:, 0] - A[:, 0]) ** 2 + (D[:, 1] - A[:, 1]) ** 2
for name, pred in (("contains the largest point index", np.maximum(b, c) > np.maximum(a, d)),
                   ("contains the smallest point index", np.minimum(b, c) < np.minimum(a, d)),
                   ("contains the largest |p|^2", np.maximum(h(B), h(C)) > np.maximum(h(A), h(D))),
                   ("contains the smallest |p|^2", np.minimum(h(B), h(C)) < np.minimum(h(A), h(D))),
                   ("has positive slope", dx * dy > 0), ("is axis-parallel", (dx == 0) | (dy == 0)),
                   ("is the shorter one", lb < la), ("(both diagonals equally long)", lb == la)):
    print(f"Qhull's diagonal {name:36s} {pred.mean():.3f}")
