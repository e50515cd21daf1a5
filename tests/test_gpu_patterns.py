"""On-device pattern construction (chest_patterns_build): pilot selection, slot table, Delaunay triangulation, plans.

Parity as SURVEY 7 (hard part 1, option ii) defines it for a native triangulator: the device triangulation is a VALID
Delaunay triangulation (exact integer predicates: positive orientation, symmetric adjacency by construction, no pilot
strictly inside any circumcircle, areas sum to the hull's), and it differs from scipy.spatial.Delaunay (= Qhull, what
griddata builds at src/baseline_estimators.py:65-79) only in triangles whose circumcircle passes through a fourth pilot,
i.e. inside co-circular lattice polygons where any triangulation is Delaunay.  Everywhere else the plans are the same
operator."""
import importlib

import numpy as np
import pytest
import torch
from scipy.spatial import ConvexHull, Delaunay

from oracle import cpu_ref as O
from oracle import philox as PH
from conftest import relerr

pytestmark = pytest.mark.gpu
E = importlib.import_module("channel-estimation-in-5g-network_amd.engine")


def _coords(idx, nsc):
    return (idx // nsc).astype(np.int64), (idx % nsc).astype(np.int64)


def _orient(ax, ay, bx, by, cx, cy):
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)


def _incircle_all(px, py, tri):
    """[T, P] exact int64 in-circle determinants of every pilot against every (counter-clockwise) triangle."""
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    adx, ady = px[a][:, None] - px[None, :], py[a][:, None] - py[None, :]
    bdx, bdy = px[b][:, None] - px[None, :], py[b][:, None] - py[None, :]
    cdx, cdy = px[c][:, None] - px[None, :], py[c][:, None] - py[None, :]
    return ((adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) - (bdx * bdx + bdy * bdy) * (adx * cdy - cdx * ady)
            + (cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady))


def _check_triangulation(idx, nsc, tri):
    px, py = _coords(idx, nsc)
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    area2 = _orient(px[a], py[a], px[b], py[b], px[c], py[c])
    assert (area2 > 0).all(), "triangles must be counter-clockwise with positive area"
    assert area2.sum() == int(round(2 * ConvexHull(np.column_stack([px, py])).volume)), "triangles must tile the convex hull"
    assert len(np.unique(tri)) == len(idx), "every pilot is a vertex"
    inc = _incircle_all(px, py, tri)
    assert (inc <= 0).all(), "a pilot lies strictly inside a circumcircle: not Delaunay"
    return inc


def _plan_matrix(eng, slot, P):
    pi, pw = eng.get_plan(slot)
    G = pi.shape[1]
    A = np.zeros((G, P))
    for c in range(3):
        np.add.at(A, (np.arange(G), pi[c].astype(np.int64)), pw[c].astype(np.float64))
    return A, pi, pw


def _scipy_plan(idx, S, nsc):
    px, py = _coords(idx, nsc)
    tri = Delaunay(np.column_stack([px, py]))
    ss, kk = np.meshgrid(np.arange(S), np.arange(nsc), indexing="ij")
    xi = np.column_stack([ss.ravel(), kk.ravel()])
    sx = tri.find_simplex(xi)
    return tri, sx


@pytest.mark.parametrize("U,densities,per_density", [(72, (0.02, 0.1, 0.3, 0.5), 12), (600, (0.01, 0.05, 0.1, 0.2, 0.4), 12)])
def test_device_delaunay_is_valid_and_equals_qhull_outside_cocircular_polygons(U, densities, per_density):
    geom = E.Geometry(useful_subcarriers=U)
    eng = E.Engine(geom, 0)
    S, nsc, G = geom.num_symbols, geom.num_sc, geom.grid
    rs = np.random.RandomState(1234 + U)
    for dens in densities:
        P = max(3, int(G * dens))
        pils = np.stack([np.sort(rs.permutation(G)[:P]) for _ in range(per_density)]).astype(np.int32)
        slots, simp, cnt = eng.build_patterns(0, [P] * per_density, pilot_idx=pils, method="linear", return_simplices=True)
        eng.check_status()
        simp, cnt = simp.cpu().numpy(), cnt.cpu().numpy()
        for j in range(per_density):
            idx = pils[j].astype(np.int64)
            tri = simp[j, :cnt[j]].astype(np.int64)
            inc = _check_triangulation(idx, nsc, tri)
            ref, sx = _scipy_plan(idx, S, nsc)
            assert cnt[j] == len(ref.simplices)
            # triangles that are not Qhull's must be degenerate: a fourth pilot exactly on their circumcircle
            ref_set = {tuple(sorted(t)) for t in ref.simplices.tolist()}
            for t in range(len(tri)):
                if tuple(sorted(tri[t].tolist())) not in ref_set:
                    assert (inc[t] == 0).sum() >= 4, "a non-degenerate triangle differs from Qhull's"
            # plans: the same operator wherever the element's Qhull simplex is also a device triangle
            A, pi, pw = _plan_matrix(eng, int(slots[j]), P)
            dev_set = {tuple(sorted(t)) for t in tri.tolist()}
            inside = sx >= 0
            assert np.array_equal(A.sum(1) > 0.5, inside), "hull membership differs"
            assert np.allclose(A[inside].sum(1), 1.0, atol=1e-5) and np.all(A[~inside] == 0.0)
            b = ref.transform[sx[inside], :2]
            bc = np.einsum("nij,nj->ni", b, np.column_stack(np.unravel_index(np.nonzero(inside)[0], (S, nsc))) - ref.transform[sx[inside], 2])
            w_ref = np.column_stack([bc, 1 - bc.sum(1)])
            A_ref = np.zeros((inside.sum(), P))
            for c in range(3):
                np.add.at(A_ref, (np.arange(inside.sum()), ref.simplices[sx[inside], c]), w_ref[:, c])
            same_tri = np.array([tuple(sorted(ref.simplices[q].tolist())) in dev_set for q in sx[inside]])
            assert same_tri.mean() > 0.6                           # the comparison below is not vacuous: most of the grid is non-degenerate
            assert np.max(np.abs(A[inside][same_tri] - A_ref[same_tri])) < 2e-6
            # ... and EVERYWHERE inside the hull -- co-circular polygons included, where the device triangle is not Qhull's -- the
            # three ordinals are a device triangle that contains the element and the three weights are its barycentric
            # coordinates (float64 from the exact integer edge functions; the plan stores float32 quotients)
            assert (pw[:, inside] >= -1e-6).all()
            px, py = _coords(idx, nsc)
            gi = np.nonzero(inside)[0]
            es, ek = np.divmod(gi, nsc)
            v = pi[:, gi].astype(np.int64)                                      # [3, n] pilot ordinals
            assert all(tuple(sorted(c)) in dev_set for c in v.T.tolist()), "a plan entry names three pilots that are not a device triangle"
            o0 = _orient(px[v[1]], py[v[1]], px[v[2]], py[v[2]], es, ek)
            o1 = _orient(px[v[2]], py[v[2]], px[v[0]], py[v[0]], es, ek)
            o2 = _orient(px[v[0]], py[v[0]], px[v[1]], py[v[1]], es, ek)
            assert (o0 >= 0).all() and (o1 >= 0).all() and (o2 >= 0).all(), "an element lies outside the triangle its plan entry names"
            den = (o0 + o1 + o2).astype(np.float64)
            assert np.max(np.abs(pw[:, gi] - np.stack([o0, o1, o2]) / den)) < 2e-6
            # affine reproduction (what the two checks above imply; float32 weights times coordinates up to nsc)
            assert np.max(np.abs(A[inside] @ px - es)) < 2e-6 * S + 1e-6 and np.max(np.abs(A[inside] @ py - ek)) < 2e-6 * nsc + 1e-6
            # slot table: pilot ordinal at pilots, P + data ordinal elsewhere
            want = np.empty(G, dtype=np.int64)
            mask = np.zeros(G, dtype=bool); mask[idx] = True
            want[mask] = np.arange(P); want[~mask] = P + np.arange(G - P)
            got = eng.run_frames(E.make_descs("EPA", 5.0, 10.0, int(slots[j]), n=1), seed=1, outputs=("pilot_mask",))["pilot_mask"][0].cpu().numpy().reshape(-1)
            assert np.array_equal(got.astype(bool), mask)


def test_device_drawn_pilots_match_the_oracle_stream_and_are_uniform():
    geom = E.Geometry()
    eng = E.Engine(geom, 0)
    G = geom.grid
    ids = np.array([0, 1, 2, 7, 2**40 + 5, 123456789], dtype=np.uint64)
    Ps = [838, 419, 83, 838, 167, 4000]
    slots = eng.build_patterns(10, Ps, seed=42, pattern_ids=ids, method="linear")
    eng.check_status()
    for s, pid, P in zip(slots, ids, Ps):
        got = eng.get_pilots(int(s))
        assert np.array_equal(got, PH.native_pilot_pattern(42, int(pid), G, P))
        assert len(np.unique(got)) == P and got.min() >= 0 and got.max() < G
    # marginal inclusion frequency over many patterns is P/G for every element (binomial tolerance)
    n = 2000
    sl = eng.build_patterns(100, [838] * n, seed=7, pattern_ids=np.arange(n, dtype=np.uint64), method=None)
    hits = np.zeros(G)
    for s in sl[::10]:
        hits[eng.get_pilots(int(s))] += 1
    p = 838 / G
    assert abs(hits.mean() / (n / 10) - p) < 1e-12
    assert hits.max() < (n / 10) * p + 6 * np.sqrt((n / 10) * p * (1 - p)) + 1


def test_native_nearest_plan_picks_a_closest_pilot_with_the_documented_tie_rule():
    geom = E.Geometry()
    eng = E.Engine(geom, 0)
    S, nsc, G = geom.num_symbols, geom.num_sc, geom.grid
    rs = np.random.RandomState(5)
    for dens in (0.01, 0.1):
        P = int(G * dens)
        idx = np.sort(rs.permutation(G)[:P]).astype(np.int32)
        slot = int(eng.build_patterns(0, [P], pilot_idx=idx[None], method="nearest")[0])
        pi, pw = eng.get_plan(slot)
        assert np.all(pw[0] == 1.0) and np.all(pw[1] == 0.0) and np.all(pi[0] == pi[1]) and np.all(pi[0] == pi[2])
        px, py = _coords(idx.astype(np.int64), nsc)
        ss, kk = np.divmod(np.arange(G), nsc)
        d2 = (ss[:, None] - px[None, :]) ** 2 + (kk[:, None] - py[None, :]) ** 2
        assert np.array_equal(pi[0].astype(np.int64), np.argmin(d2, axis=1))       # first minimum = smallest ordinal among ties
        # agrees with scipy's cKDTree wherever the nearest pilot is unique
        from scipy.spatial import cKDTree
        _, nn = cKDTree(np.column_stack([px, py])).query(np.column_stack([ss, kk]))
        unique = (d2 == d2.min(1, keepdims=True)).sum(1) == 1
        assert np.array_equal(nn[unique], pi[0].astype(np.int64)[unique]) and 0.7 < unique.mean() < 1.0


def test_rebuilding_a_slot_range_reuses_its_storage_and_frames_follow_the_new_patterns():
    geom = E.Geometry(useful_subcarriers=72)
    eng = E.Engine(geom, 0)
    G = geom.grid
    B = 64
    ids0 = np.arange(B, dtype=np.uint64)
    eng.build_patterns(0, [99] * B, seed=3, pattern_ids=ids0, method="linear")
    d = E.make_descs("EVA", 70.0, 10.0, np.arange(B, dtype=np.int32), n=B)
    a = eng.run_frames(d, seed=9, outputs=("pilot_mask", "H_ls"))
    eng.build_patterns(0, [99] * B, seed=3, pattern_ids=ids0 + 1000, method="linear")      # same slots, new patterns, in place
    b = eng.run_frames(d, seed=9, outputs=("pilot_mask", "H_ls"))
    eng.build_patterns(0, [99] * B, seed=3, pattern_ids=ids0, method="linear")
    c = eng.run_frames(d, seed=9, outputs=("pilot_mask", "H_ls"))
    for j in (0, 17, 63):
        assert np.array_equal(np.nonzero(a["pilot_mask"][j].cpu().numpy().reshape(-1))[0], PH.native_pilot_pattern(3, j, G, 99))
        assert np.array_equal(np.nonzero(b["pilot_mask"][j].cpu().numpy().reshape(-1))[0], PH.native_pilot_pattern(3, j + 1000, G, 99))
    assert torch.equal(a["H_ls"], c["H_ls"]) and not torch.equal(a["H_ls"], b["H_ls"])


@pytest.mark.parametrize("U,dens", [(72, 0.10), (72, 0.03), (600, 0.10), (600, 0.05)])
def test_device_clough_tocher_operator_equals_scipy_on_qhulls_triangulation(U, dens):
    """'cubic' natively: chest_cubic_operator builds the dense Clough-Tocher operator on the device (direct float64 solve of
    the gradient system, patch coefficients per resource element).  On Qhull's triangulation it must equal
    scipy.interpolate.CloughTocher2DInterpolator applied to the identity -- what griddata(method='cubic') does at
    src/baseline_estimators.py:65-79 -- within SciPy's own gradient-solver tolerance (tol = 1e-6)."""
    import time
    IP = importlib.import_module("channel-estimation-in-5g-network_amd.interp_plan")
    geom = E.Geometry(useful_subcarriers=U)
    eng = E.Engine(geom, 0)
    S, nsc, G = geom.num_symbols, geom.num_sc, geom.grid
    P = int(G * dens)
    idx = np.sort(np.random.RandomState(77 + U).permutation(G)[:P])
    ps, pk = np.unravel_index(idx, (S, nsc))
    simp, nbr = IP.delaunay_with_neighbors(ps, pk)
    slot = eng.pattern_slot(idx, "linear", plan=simp)
    A = eng.cubic_operator_from_triangulation(slot, simp, nbr)
    dt = 1e9
    for _ in range(3):                                            # best of three: the host side of a shared box stalls now and then
        t0 = time.perf_counter()
        A = eng.cubic_operator_from_triangulation(slot, simp, nbr)
        torch.cuda.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    ref = IP.cubic_operator(ps, pk, S, nsc)                       # SciPy's interpolator on the identity (host, ~1 s at full size)
    got = A.cpu().numpy()
    assert got.shape == ref.shape == (G, P)
    assert np.max(np.abs(got - ref)) < 3e-6                       # SciPy's default tol 1e-6 shows as ~6e-7 in the operator
    y = np.random.RandomState(1).randn(P) + 1j * np.random.RandomState(2).randn(P)
    assert np.max(np.abs(got @ y - ref @ y)) / np.max(np.abs(ref @ y)) < 3e-6
    # exact interpolation at the pilots, zero outside the hull, rows sum to one inside (the patch reproduces constants)
    assert np.max(np.abs(got[idx] - np.eye(P))) < 1e-6
    inside = np.abs(ref).sum(1) > 0
    assert np.all(got[~inside] == 0.0) and np.max(np.abs(got[inside].sum(1) - 1.0)) < 1e-5
    assert dt < (0.05 if U == 600 else 0.02), dt                  # the host route (SciPy on the identity) needs ~1 s at full size


def test_clough_tocher_on_the_native_triangulation_is_a_valid_c1_interpolant():
    """The same operator built on the device's own Delaunay triangulation: exact at the pilots, reproduces linear functions
    (Clough-Tocher has cubic precision and the gradient estimate is exact for linear data), and differs from the Qhull-based
    operator only moderately (the triangulations differ inside co-circular lattice polygons, and the global gradient solve
    spreads that)."""
    IP = importlib.import_module("channel-estimation-in-5g-network_amd.interp_plan")
    geom = E.Geometry()
    eng = E.Engine(geom, 0)
    S, nsc, G = geom.num_symbols, geom.num_sc, geom.grid
    P = 838
    slots, simp, cnt = eng.build_patterns(0, [P], seed=11, pattern_ids=np.array([5], dtype=np.uint64), method="linear", return_simplices=True)
    n = int(cnt[0])
    A = eng.cubic_operator_from_triangulation(int(slots[0]), simp[0, :n], eng._last_neighbors[0, :n]).cpu().numpy().astype(np.float64)
    idx = eng.get_pilots(int(slots[0])).astype(np.int64)
    ps, pk = np.unravel_index(idx, (S, nsc))
    assert np.max(np.abs(A[idx] - np.eye(P))) < 1e-6
    inside = np.abs(A).sum(1) > 0
    ss, kk = np.divmod(np.arange(G), nsc)
    lin = 0.3 + 0.07 * ps - 0.002 * pk
    assert np.max(np.abs(A[inside] @ lin - (0.3 + 0.07 * ss - 0.002 * kk)[inside])) < 1e-5
    ref = IP.cubic_operator(ps, pk, S, nsc).astype(np.float64)
    y = np.cos(0.05 * pk) * np.exp(-0.1 * ps)                     # smooth data: both interpolants are close to it and to each other
    assert np.max(np.abs(A @ y - ref @ y)) < 0.05 and np.array_equal(inside, np.abs(ref).sum(1) > 0)


def test_compact_plans_give_the_same_frames_as_full_plans():
    """chest_patterns_build method 2 ('linear', compact plan: a u16 triangle id per element + a 16-B record per triangle whose
    edge functions the frame kernel evaluates itself): the expanded plan equals the full one (same ordinals, weights within
    float rounding), frames run through it equal the frames run through full plans (synthesis tensors bit for bit -- the
    plan is not involved --, estimates within 1e-6), the stand-alone estimators refuse such a slot and a context refuses to
    mix the two formats in one launch."""
    N = importlib.import_module("channel-estimation-in-5g-network_amd._native")
    names = ["tx", "rx", "H", "H_ls", "H_mmse", "pilot_symbols", "pilot_mask", "metrics", "frame_status"]
    # (the last case has an odd number of used bins below DC: the generic instantiation of the frame kernel)
    for T, R, ct, Ps, U in ((2, 2, "EVA", [838, 419, 83, 838, 167, 3], 600), (1, 1, "EPA", [838, 500, 25], 600), (4, 4, "ETU", [838, 100], 600),
                            (2, 2, "ETU", [99, 30], 70)):
        geom = E.Geometry(num_tx=T, num_rx=R, useful_subcarriers=U)
        ids = np.arange(len(Ps), dtype=np.uint64) * 7 + 3
        outs, plans = [], []
        for method in ("linear", "linear_compact"):
            eng = E.Engine(geom, 0)
            assert eng.compact_plans_supported
            slots = eng.build_patterns(0, Ps, seed=99, pattern_ids=ids, method=method)
            descs = E.make_descs(ct, 70.0, 12.0, np.asarray(slots, np.int32), n=len(Ps))
            outs.append({k: v.cpu().numpy() for k, v in eng.run_frames(descs, seed=5, frame_offset=1000, outputs=names).items()})
            plans.append([eng.get_plan(int(sl)) for sl in slots])
            if method == "linear_compact":
                with pytest.raises(N.ChestError):
                    eng.estimate(np.zeros((geom.num_symbols, R, T, geom.num_sc), np.complex64), outs[-1]["pilot_symbols"][0, :Ps[0]], int(slots[0]))
                eng.pattern_slot(eng.get_pilots(int(slots[0])), "linear")        # a full plan joins the compact ones
                with pytest.raises(N.ChestError):
                    eng.run_frames(descs, seed=5, outputs=("H_ls",))
                eng.run_frames(descs, seed=5, outputs=("rx",))                    # no estimates requested: no plan is read
        a, b = outs
        assert not a["frame_status"].any() and not b["frame_status"].any()
        for k in ("tx", "rx", "H", "pilot_symbols", "pilot_mask"):
            assert np.array_equal(a[k], b[k]), k
        for k in ("H_ls", "H_mmse"):
            assert np.max(np.abs(a[k] - b[k])) <= 1e-6 * np.max(np.abs(a[k])), k
        assert np.allclose(a["metrics"], b["metrics"], rtol=1e-5, atol=0)
        for (ia, wa), (ib, wb) in zip(*plans):
            inside = wa.sum(0) > 0.5
            assert np.array_equal(inside, wb.sum(0) > 0.5)
            assert np.array_equal(ia[:, inside], ib[:, inside]) and np.max(np.abs(wa - wb)) < 1e-6


def _interpolate_on(simplices, px, py, h_p, S, nsc):
    """Piecewise-linear interpolation of pilot values on a GIVEN triangulation (float64 barycentric weights from the exact
    integer edge functions; 0 outside the hull) -- griddata('linear') with scipy's Qhull triangulation replaced by `simplices`.
    An element on a shared edge gets the same value from either triangle."""
    G = S * nsc
    es, ek = np.divmod(np.arange(G), nsc)
    a, b, c = simplices[:, 0], simplices[:, 1], simplices[:, 2]
    out = np.zeros(G, dtype=complex)
    done = np.zeros(G, dtype=bool)
    for lo in range(0, len(simplices), 256):                                   # [G, 256] blocks of edge functions
        sl = slice(lo, lo + 256)
        o0 = _orient(px[b[sl]][None], py[b[sl]][None], px[c[sl]][None], py[c[sl]][None], es[:, None], ek[:, None])
        o1 = _orient(px[c[sl]][None], py[c[sl]][None], px[a[sl]][None], py[a[sl]][None], es[:, None], ek[:, None])
        o2 = _orient(px[a[sl]][None], py[a[sl]][None], px[b[sl]][None], py[b[sl]][None], es[:, None], ek[:, None])
        hit = (o0 >= 0) & (o1 >= 0) & (o2 >= 0)
        g, t = np.nonzero(hit & ~done[:, None])
        g, first = np.unique(g, return_index=True)
        t = t[first]
        den = (o0[g, t] + o1[g, t] + o2[g, t]).astype(np.float64)
        out[g] = (o0[g, t] * h_p[a[sl][t]] + o1[g, t] * h_p[b[sl][t]] + o2[g, t] * h_p[c[sl][t]]) / den
        done[g] = True
    return out.reshape(S, nsc)


@pytest.mark.parametrize("method", ["linear", "linear_compact"])
def test_frames_on_device_built_patterns_equal_the_oracle_on_the_device_triangulation(method):
    """End to end through plans the DEVICE triangulated (full and compact format, P = 838 / 419 / 83 drawn on the device):
    H_ls and H_mmse of the fused pipeline against the oracle's LS / MMSE at the pilots interpolated on the triangulation the
    device reports (return_simplices) -- so a weight error confined to co-circular polygons, where the device's triangles
    are not Qhull's and the Qhull-based goldens cannot see it, fails here."""
    geom = E.Geometry()
    S, nsc, G = geom.num_symbols, geom.num_sc, geom.grid
    Ps = [838, 419, 83]
    ids = np.array([11, 12, 13], dtype=np.uint64)
    eng_t = E.Engine(geom, 0)                                  # the triangulation a pattern gets does not depend on the plan format
    _, simp, cnt = eng_t.build_patterns(0, Ps, seed=21, pattern_ids=ids, method="linear", return_simplices=True)
    simp, cnt = simp.cpu().numpy().astype(np.int64), cnt.cpu().numpy()
    eng = E.Engine(geom, 0)
    slots = eng.build_patterns(0, Ps, seed=21, pattern_ids=ids, method=method)
    cts, fds, snrs = ["EVA", "ETU", "EPA"], [111.1, 300.0, 5.0], [10.0, 20.0, 0.0]
    descs = E.make_descs(cts, fds, snrs, np.asarray(slots, np.int32))
    out = {k: v.cpu().numpy() for k, v in eng.run_frames(descs, seed=31, frame_offset=7, outputs=("rx", "H_ls", "H_mmse", "pilot_symbols")).items()}
    for b, P in enumerate(Ps):
        pil = eng.get_pilots(int(slots[b])).astype(np.int64)
        assert np.array_equal(pil, eng_t.get_pilots(b)) and np.array_equal(pil, PH.native_pilot_pattern(21, int(ids[b]), G, P))
        px, py = _coords(pil, nsc)
        tri = simp[b, :cnt[b]]
        _check_triangulation(pil, nsc, tri)
        mask = np.zeros(G, dtype=bool); mask[pil] = True
        mask = mask.reshape(S, nsc)
        x = out["pilot_symbols"][b, :P].astype(complex)
        rx = out["rx"][b].astype(complex)                      # (S, R, nsc): the estimators see rx repeated over tx (quirk Q9)
        for r in range(geom.num_rx):
            h_ls = rx[:, r, :][mask] / (x + 1e-12)
            want_ls = _interpolate_on(tri, px, py, h_ls, S, nsc)
            want_mm = _interpolate_on(tri, px, py, O.mmse_at_pilots(h_ls, snrs[b], dense=False), S, nsc)
            for t in range(geom.num_tx):
                assert relerr(out["H_ls"][b][:, r, t, :], want_ls) < 1e-5, (method, P, r, t)
                assert relerr(out["H_mmse"][b][:, r, t, :], want_mm) < 1e-5, (method, P, r, t)


def test_flip_round_limit_raises_a_pattern_status():
    """The Delaunay flips of chest_patterns_build stop at a round limit (4 * triangles + 64, never reached); a pattern that
    runs into it must not pass for a good one: status 3.  Forced here with chest_tune("pattern_flip_cap")."""
    N = importlib.import_module("channel-estimation-in-5g-network_amd._native")
    geom = E.Geometry()
    for P in (838, 4000):                                       # barrier-free schedule / workgroup-synchronous schedule (> 2 048 triangles)
        eng = E.Engine(geom, 0)
        eng.tune("pattern_flip_cap", 2)
        eng.build_patterns(0, [P] * 3, seed=1, pattern_ids=np.arange(3, dtype=np.uint64), method="linear")
        with pytest.raises(N.ChestError, match="round limit"):
            eng.check_status()
        eng.tune("pattern_flip_cap", 0)
        eng.build_patterns(0, [P] * 3, seed=1, pattern_ids=np.arange(3, dtype=np.uint64), method="linear")
        eng.check_status()


@pytest.mark.parametrize("U", [72, 600])
def test_structured_and_tiny_patterns_triangulate_validly(U):
    """Patterns a uniform draw never produces: one or two occupied rows, a regular lattice (every adjacent triangle pair
    co-circular), a sheared lattice on every other row, one pilot per row (every strip empty: the pocket scan's shared chain
    edges), the two outer columns only, a cluster, a single triangle, three collinear pilots, and tiny random sets -- the
    device triangulation must be a valid Delaunay triangulation with Qhull's triangle count and hull (tools/stress_patterns.py
    runs the same checks on a few hundred patterns)."""
    geom = E.Geometry(useful_subcarriers=U)
    eng = E.Engine(geom, 0)
    S, nsc, G = geom.num_symbols, geom.num_sc, geom.grid
    rs = np.random.RandomState(99 + U)
    c, w = nsc // 2, min(20, nsc // 4)
    pats = [np.arange(0, nsc, 7),
            np.concatenate([np.arange(3, nsc, 11), (S - 1) * nsc + np.arange(0, nsc, 5)]),
            np.asarray([s * nsc + k for s in range(S) for k in range(0, nsc, 5)]),
            np.asarray([s * nsc + k for s in range(0, S, 2) for k in range(s % 4, nsc, 4)]),
            np.asarray([s * nsc + (17 * s) % nsc for s in range(S)]),
            np.asarray([s * nsc + k for s in range(S) for k in (0, nsc - 1)]),
            np.unique(np.asarray([rs.randint(S) * nsc + c + rs.randint(-w, w + 1) for _ in range(200)])),
            np.asarray([0, nsc - 1, (S - 1) * nsc + nsc // 2]),
            np.asarray([5, 6, 7])]
    pats += [rs.permutation(G)[:P] for P in (3, 4, 5, 7, 10, 14, 20) for _ in range(3)]
    pats = [np.sort(np.unique(p)).astype(np.int32) for p in pats]
    for lo in range(0, len(pats), 8):
        chunk = pats[lo:lo + 8]
        idx = np.zeros((len(chunk), max(len(p) for p in chunk)), np.int32)
        for j, p in enumerate(chunk):
            idx[j, :len(p)] = p
        slots, simp, cnt = eng.build_patterns(lo, [len(p) for p in chunk], pilot_idx=idx, method="linear", return_simplices=True)
        eng.check_status()
        simp, cnt = simp.cpu().numpy(), cnt.cpu().numpy()
        for j, p in enumerate(chunk):
            p64 = p.astype(np.int64)
            px, py = _coords(p64, nsc)
            if np.all(_orient(px[0], py[0], px[1], py[1], px, py) == 0):           # collinear: no triangle, everything outside
                assert cnt[j] == 0
                A, _, _ = _plan_matrix(eng, int(slots[j]), len(p))
                assert not A.any()
                continue
            tri = simp[j, :cnt[j]].astype(np.int64)
            _check_triangulation(p64, nsc, tri)
            ref, sx = _scipy_plan(p64, S, nsc)
            assert cnt[j] == len(ref.simplices)
            A, pi, pw = _plan_matrix(eng, int(slots[j]), len(p))
            assert np.array_equal(A.sum(1) > 0.5, sx >= 0), "hull membership differs"


@pytest.mark.parametrize("P", [838, 419])
def test_device_plan_divergence_from_griddata_is_measured_and_bounded(P):
    """The batch path's relaxation, as a NUMBER: 64 device-drawn patterns, the same frames (identical tx / H / rx, checked bit for
    bit) once through the device triangulator's compact plans and once through plans located in Qhull's simplices -- the
    triangulation griddata('linear') uses (src/baseline_estimators.py:65-79).  Inside co-circular lattice polygons the two valid
    Delaunay triangulations interpolate differently: a few per cent of the resource elements differ by more than north_star's
    1e-5, the LS estimate by a few per cent in relative L2, a frame's LS MSE by ~1e-3 relative (no bias: the batch MSE agrees to
    ~1e-4).  bench.py reports the same record as `per_frame_patterns.plan_parity`; the shared-pattern headline runs on Qhull plans
    and has no such term."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import plan_parity
    r = plan_parity.measure(P=P, patterns=64)
    print(r)
    assert r["patterns"] >= 50 and 0.85 < r["shared_triangles"] < 1.0
    assert 0.0 < r["H_ls"]["re_frac_gt_1e5"] < 0.05                 # measured 0.02-0.03
    assert r["H_ls"]["rel_l2"] < 0.05                               # measured 0.026-0.03
    assert r["H_mmse"]["re_frac_gt_1e5"] < 0.05 and r["H_mmse"]["rel_l2"] < 0.05
    assert r["mse_rel_delta_ls"]["mean"] < 3e-3 and r["mse_rel_delta_ls"]["max"] < 2e-2
    assert r["mse_rel_delta_ls"]["batch"] < 1e-3                    # and it does not bias the batch statistic


def test_device_nearest_plan_divergence_from_ckdtree_is_measured_and_bounded():
    """'nearest' (griddata(method='nearest') = cKDTree, src/baseline_estimators.py:65-79): SURVEY 7 counted 14.6 % of the resource
    elements with two or more pilots at exactly the minimum distance.  The device search (batch path) takes the smallest pilot ordinal,
    cKDTree whatever its traversal meets first.  Same frames through both plans (missing 5): both always pick A nearest
    pilot, they differ at tied elements only, and the divergence is a recorded number.  The drop-in LSEstimator('nearest') uses the
    cKDTree plan itself and has no such term."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import plan_parity
    r = plan_parity.measure_nearest(P=838, patterns=16)
    print(r)
    assert 0.08 < r["tie_frac"] < 0.25                               # SURVEY: 14.6 %
    assert r["picks_differing_at_untied_elements"] == 0
    assert 0.0 < r["pick_differs"] <= r["tie_frac"]
    assert r["H_ls"]["re_frac_gt_1e5"] <= r["pick_differs"] * 1.0001 and r["H_ls"]["rel_l2"] < 0.5
    assert r["mse_rel_delta_ls"]["batch"] < 2e-2
