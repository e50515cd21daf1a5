"""Interpolation plans: the geometry-only part of ``scipy.interpolate.griddata`` done once per
pilot pattern on the host, so that the GPU applies it as a 3-point gather.

``LSEstimator.interpolate_channel`` (reference src/baseline_estimators.py:44-81) calls
``griddata(points=(sym, sc) integer coordinates, xi=all (sym, sc) row-major, method, fill_value=0)``
twice per antenna pair.  For ``'linear'`` that is: Qhull Delaunay triangulation of the pilots
(options "Qbb Qc Qz Q12", no rescaling), locate each grid point's simplex, barycentric weights,
0 outside the convex hull.  The pilots sit on an integer lattice, so the triangulation is massively
degenerate and only Qhull's own tie-breaking reproduces the reference; the plan is therefore
extracted from ``scipy.spatial.Delaunay`` itself.  ``'nearest'`` is a cKDTree query
(``NearestNDInterpolator``), a plan with weights (1, 0, 0).

A plan is ``(idx uint16[3, G], w float32[3, G])`` with G = S*nsc in row-major grid order.

For ``'linear'`` the engine's default is to take only the triangulation from the host
(``linear_triangulation``: Qhull, ~3 ms per full-size pattern) and to do the point location and the
barycentric weights on the GPU (``chest_pattern_set_triangulation``); ``linear_plan`` -- SciPy's own
``find_simplex`` + ``transform``, ~45 ms -- stays as the cross-check and for callers that want the plan
on the host.
"""
from __future__ import annotations

import hashlib

import numpy as np

_CACHE = {}
_CACHE_MAX = 256


def _grid_points(S, nsc):
    ss, kk = np.meshgrid(np.arange(S), np.arange(nsc), indexing="ij")
    return np.column_stack([ss.ravel(), kk.ravel()]).astype(np.float64)


def linear_triangulation(pilot_sym, pilot_sc):
    """``scipy.spatial.Delaunay(points).simplices`` (int32 [n, 3] pilot ordinals) of the pilot positions, i.e. the
    triangulation griddata's LinearNDInterpolator builds.  Raises scipy's QhullError for degenerate pilot sets
    (fewer than 3 pilots, all collinear), like the reference does."""
    from scipy.spatial import Delaunay
    pts = np.column_stack([np.asarray(pilot_sym), np.asarray(pilot_sc)]).astype(np.float64)
    return np.ascontiguousarray(Delaunay(pts).simplices, dtype=np.int32)


def delaunay_with_neighbors(pilot_sym, pilot_sc):
    """(simplices, neighbors) int32 [n, 3] of scipy.spatial.Delaunay(pilot positions): what CloughTocher2DInterpolator works on."""
    from scipy.spatial import Delaunay
    tri = Delaunay(np.column_stack([np.asarray(pilot_sym), np.asarray(pilot_sc)]).astype(np.float64))
    return np.ascontiguousarray(tri.simplices, dtype=np.int32), np.ascontiguousarray(tri.neighbors, dtype=np.int32)


def linear_plan(pilot_sym, pilot_sc, S, nsc):
    """Plan equal to griddata(..., method='linear', fill_value=0.0).  Raises scipy's QhullError for
    degenerate pilot sets (fewer than 3 pilots, all collinear), like the reference does."""
    from scipy.spatial import Delaunay
    pts = np.column_stack([np.asarray(pilot_sym), np.asarray(pilot_sc)]).astype(np.float64)
    tri = Delaunay(pts)
    xi = _grid_points(S, nsc)
    simp = tri.find_simplex(xi)
    inside = simp >= 0
    sj = np.where(inside, simp, 0)
    Tm = tri.transform[sj]                                  # (G, 3, 2)
    bary = np.einsum("gij,gj->gi", Tm[:, :2, :], xi - Tm[:, 2, :])
    w = np.concatenate([bary, 1.0 - bary.sum(axis=1, keepdims=True)], axis=1)
    w[~inside] = 0.0
    idx = tri.simplices[sj].astype(np.int64)
    idx[~inside] = 0
    # a simplex may be flagged degenerate by Qhull (nan transform); griddata yields nan there
    return np.ascontiguousarray(idx.T.astype(np.uint16)), np.ascontiguousarray(w.T.astype(np.float32))


def nearest_plan(pilot_sym, pilot_sc, S, nsc):
    """Plan equal to griddata(..., method='nearest')."""
    from scipy.spatial import cKDTree
    pts = np.column_stack([np.asarray(pilot_sym), np.asarray(pilot_sc)]).astype(np.float64)
    _, i = cKDTree(pts).query(_grid_points(S, nsc))
    G = S * nsc
    idx = np.empty((3, G), dtype=np.uint16)
    idx[:] = i.astype(np.uint16)
    w = np.zeros((3, G), dtype=np.float32)
    w[0] = 1.0
    return idx, w


_DENSE_CACHE = {}
_DENSE_CACHE_MAX = 4


def cubic_operator(pilot_sym, pilot_sc, S, nsc):
    """Dense operator A [G, P] (float32) with griddata(points, y, xi, method='cubic', fill_value=0) == A @ y.

    'cubic' = scipy's CloughTocher2DInterpolator: Qhull Delaunay (same triangulation as 'linear'), global gradient
    estimation at the pilots (iterative, tol 1e-6), then a C1 piecewise-cubic Bezier patch per triangle.  Every step is
    linear in the pilot values, so the interpolator applied to the identity gives the operator (measured 7e-7..1.5e-6
    relative against griddata on real data: the gradient solver's own tolerance).  About 1-3 s per full-size pattern on
    one core.  HOST CROSS-CHECK ONLY: the product builds the operator on the device (Engine.cubic_operator ->
    chest_cubic_operator); tests compare the two."""
    from scipy.interpolate import CloughTocher2DInterpolator
    ps = np.ascontiguousarray(pilot_sym, dtype=np.int64)
    pk = np.ascontiguousarray(pilot_sc, dtype=np.int64)
    key = (S, nsc, hashlib.blake2b(ps.tobytes() + pk.tobytes(), digest_size=16).digest())
    hit = _DENSE_CACHE.get(key)
    if hit is None:
        pts = np.column_stack([ps, pk]).astype(np.float64)
        ct = CloughTocher2DInterpolator(pts, np.eye(len(ps)), fill_value=0.0)
        hit = np.ascontiguousarray(ct(_grid_points(S, nsc)), dtype=np.float32)
        if len(_DENSE_CACHE) >= _DENSE_CACHE_MAX:
            _DENSE_CACHE.pop(next(iter(_DENSE_CACHE)))
        _DENSE_CACHE[key] = hit
    return hit


def build_plan(pilot_sym, pilot_sc, S, nsc, method="linear"):
    """Cached 3-point plan for one pattern ('linear' / 'nearest').  'cubic' is a dense operator: cubic_operator()."""
    if method not in ("linear", "nearest"):
        if method == "cubic":
            raise ValueError("'cubic' has no 3-point plan; use interp_plan.cubic_operator (LSEstimator does)")
        raise ValueError(f"Unknown interpolation method: {method}")
    ps = np.ascontiguousarray(pilot_sym, dtype=np.int64)
    pk = np.ascontiguousarray(pilot_sc, dtype=np.int64)
    key = (method, S, nsc, hashlib.blake2b(ps.tobytes() + pk.tobytes(), digest_size=16).digest())
    hit = _CACHE.get(key)
    if hit is None:
        hit = linear_plan(ps, pk, S, nsc) if method == "linear" else nearest_plan(ps, pk, S, nsc)
        if len(_CACHE) >= _CACHE_MAX:
            _CACHE.pop(next(iter(_CACHE)))
        _CACHE[key] = hit
    return hit


def _tri_job(args):
    pilot_idx, S, nsc = args
    ps, pk = np.unravel_index(np.asarray(pilot_idx, dtype=np.int64), (S, nsc))
    return linear_triangulation(ps, pk)


def triangulations_parallel(pilot_idx_list, S, nsc, workers=0):
    """Delaunay simplices for many patterns, optionally in a spawn-context process pool; input order."""
    jobs = [(np.asarray(p, dtype=np.int64), S, nsc) for p in pilot_idx_list]
    if workers and workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(min(workers, len(jobs))) as pool:
            return pool.map(_tri_job, jobs, chunksize=max(1, len(jobs) // (4 * workers)))
    return [_tri_job(j) for j in jobs]


def triangulations_iter(pilot_idx_list, S, nsc, workers=0):
    """Like triangulations_parallel, but yields the simplices in input order as they become available, so that the caller
    can upload pattern i (~0.2 ms of host work) while the pool is still triangulating pattern i + k (~2 ms per core)."""
    jobs = [(np.asarray(p, dtype=np.int64), S, nsc) for p in pilot_idx_list]
    if workers and workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(min(workers, len(jobs))) as pool:
            yield from pool.imap(_tri_job, jobs, chunksize=max(1, min(64, len(jobs) // (8 * workers))))
    else:
        for j in jobs:
            yield _tri_job(j)


def _plan_job(args):
    """Worker entry point of build_plans_parallel (top-level so that it pickles)."""
    pilot_idx, S, nsc, method = args
    ps, pk = np.unravel_index(np.asarray(pilot_idx, dtype=np.int64), (S, nsc))
    return linear_plan(ps, pk, S, nsc) if method == "linear" else nearest_plan(ps, pk, S, nsc)


def build_plans_parallel(pilot_idx_list, S, nsc, method="linear", workers=0):
    """Plans for many patterns; with workers > 1 in a spawn-context process pool (Qhull + point location cost ~10 ms per
    full-size pattern per core and hold the GIL).  Results are in input order and identical to build_plan's."""
    if method not in ("linear", "nearest"):
        raise ValueError(f"Unknown interpolation method: {method}")
    jobs = [(np.asarray(p, dtype=np.int64), S, nsc, method) for p in pilot_idx_list]
    if workers and workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(min(workers, len(jobs))) as pool:
            return pool.map(_plan_job, jobs, chunksize=max(1, len(jobs) // (4 * workers)))
    return [_plan_job(j) for j in jobs]


def pattern_key(pilot_idx, method):
    a = np.ascontiguousarray(pilot_idx, dtype=np.int64)
    return (method, hashlib.blake2b(a.tobytes(), digest_size=16).digest())
