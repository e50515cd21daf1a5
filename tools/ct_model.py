#!/usr/bin/env python3
"""CPU model of the on-device Clough-Tocher ('cubic') operator construction (csrc/cubic_kernels.hip follows it).

griddata(method='cubic') = scipy.interpolate.CloughTocher2DInterpolator (the reference reaches it through
src/baseline_estimators.py:65-79 with interpolation_method='cubic'): on a Delaunay triangulation,
  1. gradients at the pilots from the global curvature-minimising estimate: SciPy iterates (Gauss-Seidel, tol 1e-6) on the
     linear system   sum_j (4 e e^T / L^3) g_i + sum_j (2 e e^T / L^3) g_j = 6 sum_j (f_j - f_i) e / L^3   over the
     edge neighbours j of i, e = p_j - p_i, L = |e|; here it is assembled (2P x 2P, symmetric positive definite) and solved
     DIRECTLY, which is linear in the pilot values: gradients = Gop f with Gop = M^-1 B (2P x P);
  2. per resource element the C1 piecewise-cubic Clough-Tocher Bezier patch of its triangle (12 control points from the 3
     vertex values, the 3 vertex gradients and the neighbour centroids), evaluated at the element's barycentric
     coordinates -- linear in (f, grad f) of the three vertices: 3 + 6 coefficients per element.
So the interpolant is the dense operator  A = Eval + Egrad Gop  [G, P];  this module builds it in float64 from a
triangulation (simplices + neighbours) and is checked against SciPy's interpolator applied to the identity.
"""
import numpy as np


def edge_neighbours(simplices, npts):
    nb = [set() for _ in range(npts)]
    for a, b, c in simplices:
        nb[a].update((b, c)); nb[b].update((a, c)); nb[c].update((a, b))
    return nb


def gradient_operator(pts, simplices):
    """Gop [2P, P]: stacked (d/dx, d/dy) gradient estimates at the pilots as a linear map of the pilot values."""
    P = len(pts)
    M = np.zeros((2 * P, 2 * P))
    B = np.zeros((2 * P, P))
    nb = edge_neighbours(simplices, P)
    for i in range(P):
        for j in nb[i]:
            e = pts[j] - pts[i]
            L3 = np.hypot(e[0], e[1]) ** 3
            ee = np.outer(e, e) / L3
            M[2 * i:2 * i + 2, 2 * i:2 * i + 2] += 4 * ee
            M[2 * i:2 * i + 2, 2 * j:2 * j + 2] += 2 * ee
            B[2 * i:2 * i + 2, j] += 6 * e / L3
            B[2 * i:2 * i + 2, i] -= 6 * e / L3
    return np.linalg.solve(M, B)


def ct_coefficients(p, nbr_centroids, q):
    """The 9 linear coefficients of the Clough-Tocher patch value at point q of the triangle with vertices p[0..2]
    (float64 [3, 2]): returns (cf[3], cg[3, 2]) with value = sum_v cf[v] f_v + cg[v] . grad_v.
    nbr_centroids[k]: centroid of the triangle across the edge opposite vertex k, or None on the hull."""
    T = np.array([[p[0, 0] - p[2, 0], p[1, 0] - p[2, 0]], [p[0, 1] - p[2, 1], p[1, 1] - p[2, 1]]])
    Ti = np.linalg.inv(T)

    def bary(x):
        l = Ti @ (np.asarray(x) - p[2])
        return np.array([l[0], l[1], 1 - l[0] - l[1]])

    b = bary(q)
    e12, e23, e31 = p[1] - p[0], p[2] - p[1], p[0] - p[2]
    g = np.empty(3)
    for k in range(3):
        if nbr_centroids[k] is None:
            g[k] = -0.5
            continue
        c = bary(nbr_centroids[k])
        if k == 0:
            g[k] = (2 * c[2] + c[1] - 1) / (2 - 3 * c[2] - 3 * c[1])
        elif k == 1:
            g[k] = (2 * c[0] + c[2] - 1) / (2 - 3 * c[0] - 3 * c[2])
        else:
            g[k] = (2 * c[1] + c[0] - 1) / (2 - 3 * c[1] - 3 * c[0])
    out_f, out_g = np.zeros(3), np.zeros((3, 2))
    # evaluate the patch for each of the 9 unit inputs (it is linear in them)
    for unit in range(9):
        f = np.zeros(3); df = np.zeros((3, 2))
        if unit < 3:
            f[unit] = 1.0
        else:
            df[(unit - 3) // 2, (unit - 3) % 2] = 1.0
        f1, f2, f3 = f
        df12 = df[0] @ e12; df21 = -(df[1] @ e12)
        df23 = df[1] @ e23; df32 = -(df[2] @ e23)
        df31 = df[2] @ e31; df13 = -(df[0] @ e31)
        c3000 = f1; c2100 = (df12 + 3 * c3000) / 3; c2010 = (df13 + 3 * c3000) / 3
        c0300 = f2; c1200 = (df21 + 3 * c0300) / 3; c0210 = (df23 + 3 * c0300) / 3
        c0030 = f3; c1020 = (df31 + 3 * c0030) / 3; c0120 = (df32 + 3 * c0030) / 3
        c2001 = (c2100 + c2010 + c3000) / 3
        c0201 = (c1200 + c0300 + c0210) / 3
        c0021 = (c1020 + c0120 + c0030) / 3
        c0111 = (g[0] * (-c0300 + 3 * c0210 - 3 * c0120 + c0030) + (-c0300 + 2 * c0210 - c0120 + c0021 + c0201)) / 2
        c1011 = (g[1] * (-c0030 + 3 * c1020 - 3 * c2010 + c3000) + (-c0030 + 2 * c1020 - c2010 + c2001 + c0021)) / 2
        c1101 = (g[2] * (-c3000 + 3 * c2100 - 3 * c1200 + c0300) + (-c3000 + 2 * c2100 - c1200 + c2001 + c0201)) / 2
        c1002 = (c1101 + c1011 + c2001) / 3
        c0102 = (c1101 + c0111 + c0201) / 3
        c0012 = (c1011 + c0111 + c0021) / 3
        c0003 = (c1002 + c0102 + c0012) / 3
        mn = min(b)
        b1, b2, b3, b4 = b[0] - mn, b[1] - mn, b[2] - mn, 3 * mn
        w = (b1**3 * c3000 + 3 * b1**2 * b2 * c2100 + 3 * b1**2 * b3 * c2010 + 3 * b1**2 * b4 * c2001 + 3 * b1 * b2**2 * c1200
             + 6 * b1 * b2 * b4 * c1101 + 3 * b1 * b3**2 * c1020 + 6 * b1 * b3 * b4 * c1011 + 3 * b1 * b4**2 * c1002
             + b2**3 * c0300 + 3 * b2**2 * b3 * c0210 + 3 * b2**2 * b4 * c0201 + 3 * b2 * b3**2 * c0120 + 6 * b2 * b3 * b4 * c0111
             + 3 * b2 * b4**2 * c0102 + b3**3 * c0030 + 3 * b3**2 * b4 * c0021 + 3 * b3 * b4**2 * c0012 + b4**3 * c0003)
        if unit < 3:
            out_f[unit] = w
        else:
            out_g[(unit - 3) // 2, (unit - 3) % 2] = w
    return out_f, out_g


def ct_operator(pts, simplices, neighbors, simplex_of, xi):
    """A [len(xi), P] from a triangulation: simplices [n,3], neighbors [n,3] (-1 on the hull; neighbors[t][k] is across the
    edge opposite vertex k, scipy's convention), simplex_of [len(xi)] (-1 outside the hull)."""
    P = len(pts)
    Gop = gradient_operator(pts, simplices)
    A = np.zeros((len(xi), P))
    cent = pts[simplices].mean(axis=1)
    for n, (q, t) in enumerate(zip(xi, simplex_of)):
        if t < 0:
            continue
        v = simplices[t]
        nc = [cent[neighbors[t, k]] if neighbors[t, k] >= 0 else None for k in range(3)]
        cf, cg = ct_coefficients(pts[v].astype(float), nc, q)
        for a in range(3):
            A[n, v[a]] += cf[a]
            A[n] += cg[a, 0] * Gop[2 * v[a]] + cg[a, 1] * Gop[2 * v[a] + 1]
    return A


if __name__ == "__main__":
    from scipy.interpolate import CloughTocher2DInterpolator
    from scipy.spatial import Delaunay
    S, nsc = 14, 71
    G = S * nsc
    for dens, seed in ((0.1, 1), (0.05, 2)):
        P = int(G * dens)
        idx = np.sort(np.random.RandomState(seed).permutation(G)[:P])
        pts = np.column_stack([idx // nsc, idx % nsc]).astype(float)
        tri = Delaunay(pts)
        ss, kk = np.meshgrid(np.arange(S), np.arange(nsc), indexing="ij")
        xi = np.column_stack([ss.ravel(), kk.ravel()]).astype(float)
        A = ct_operator(pts, tri.simplices, tri.neighbors, tri.find_simplex(xi), xi)
        ref = CloughTocher2DInterpolator(tri, np.eye(P), fill_value=0.0, tol=1e-12, maxiter=4000)(xi)
        ref6 = CloughTocher2DInterpolator(tri, np.eye(P), fill_value=0.0)(xi)
        y = np.random.RandomState(9).randn(P) + 1j * np.random.RandomState(10).randn(P)
        print(f"density {dens}: max|A - scipy(tol 1e-12)| = {np.max(np.abs(A - ref)):.2e}   max|A - scipy(default tol)| = {np.max(np.abs(A - ref6)):.2e}"
              f"   on data: {np.max(np.abs(A @ y - ref6 @ y)) / np.max(np.abs(ref6 @ y)):.2e}")
