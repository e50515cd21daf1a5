"""Host-side logic of the product package (no GPU): interpolation plans, pilot patterns, tables,
descriptor packing, sharding arithmetic, the Philox restatement."""
import importlib
import json
import os

import numpy as np
import pytest
from scipy.interpolate import griddata

from conftest import GOLDEN, load_golden
from oracle import cpu_ref as O
from oracle import philox as PH

PKG = "channel-estimation-in-5g-network_amd"
ip = importlib.import_module(PKG + ".interp_plan")
S_ = importlib.import_module(PKG + ".sharding")
cs = importlib.import_module(PKG + ".channel_simulator")
E = importlib.import_module(PKG + ".engine")


@pytest.mark.parametrize("name,S,nsc", [("small_2x2_eva_d10", 14, 71), ("small_4x4_epa_d01", 14, 71), ("full_cfg2_2x2_eva", 14, 599)])
@pytest.mark.parametrize("method", ["linear", "nearest"])
def test_plan_reproduces_griddata(name, S, nsc, method):
    g = load_golden(name)
    ps, pk = np.unravel_index(g["pilot_indices"], (S, nsc))
    idx, w = ip.build_plan(ps, pk, S, nsc, method)
    assert idx.shape == (3, S * nsc) and idx.dtype == np.uint16 and w.dtype == np.float32
    v = np.random.default_rng(0).standard_normal(len(ps))
    ref = griddata(np.column_stack([ps, pk]), v, ip._grid_points(S, nsc), method=method, fill_value=0.0)
    got = (w.astype(np.float64) * v[idx.astype(int)]).sum(0)
    assert np.max(np.abs(got - ref)) < 5e-7 * max(1.0, np.max(np.abs(v)))        # float32 weights
    outside = w.sum(0) == 0
    assert np.all(ref[outside] == 0.0)
    # exact at the pilots themselves
    flat = ps * nsc + pk
    assert np.max(np.abs(got[flat] - v)) < 5e-7 * np.max(np.abs(v))


def test_plan_errors():
    with pytest.raises(ValueError):
        ip.build_plan([0, 1, 2], [0, 5, 9], 14, 71, "bogus")
    from scipy.spatial import QhullError
    with pytest.raises(QhullError):
        ip.build_plan([0, 0, 0], [1, 5, 9], 14, 71, "linear")                      # collinear pilots, like the reference


def test_pilot_pattern_is_bit_exact_with_reference_stream():
    g = load_golden("full_quickstart_seed42")
    np.random.seed(int(g["seed"]))
    pp = cs.PilotPattern(599, 14, 0.1)
    assert np.array_equal(pp.pilot_indices, g["pilot_indices"])
    assert pp.pilot_indices[:6].tolist() == [19, 23, 33, 37, 48, 61]
    assert pp.pilot_mask.sum() == 838 and pp.pilot_mask.dtype == bool
    assert np.array_equal(np.flatnonzero(pp.pilot_mask.ravel()), g["pilot_indices"])
    grid = pp.insert_pilots(np.arange(14 * 599 - 838) + 1000j, np.arange(838) * 1.0)
    assert np.array_equal(pp.extract_pilots(grid), np.arange(838) * 1.0)
    assert int(14 * 71 * 0.1) == 99 and cs.PilotPattern(71, 14, 0.1).pilot_mask.sum() == 99   # int() truncation


def test_channel_model_tables_and_ofdm_indices_match_reference_kat():
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))
    for key, v in kat["tap_tables"].items():
        ct, N = key.split("_")
        cm = cs.ChannelModel(ct.lower(), 50.0, 2e9, int(N) * 15000.0)                # type is upper-cased
        assert cm.model_type == ct and cm.delay_samples.tolist() == v["delay_samples"]
        assert np.allclose(cm.powers_linear, v["powers_linear"], rtol=0, atol=1e-16)
        assert cm.num_paths == len(v["delay_samples"])
    with pytest.raises(KeyError):
        cs.ChannelModel("XYZ", 50.0, 2e9, 15.36e6)
    for key, v in kat["used_indices"].items():
        N, U = map(int, key.split("_"))
        o = cs.OFDMSystem(cs.OFDMConfig(fft_size=N, useful_subcarriers=U))
        u = o.used_indices
        assert (int(u[0]), int(u[-1]), len(u), int(u.sum())) == (v["first"], v["last"], v["count"], v["sum"])
        assert o.dc_idx == N // 2 and o.sampling_rate == N * 15000.0
    assert cs.OFDMConfig() == cs.OFDMConfig(1024, 72, 14, 600, 15000.0) and cs.MIMOConfig() == cs.MIMOConfig(2, 2)


def test_descriptor_packing_and_geometry():
    d = E.make_descs(["EPA", "eva", "ETU"], [10, 50, 100], 12.5, 3)
    assert d.dtype.itemsize == 16 and d["channel_type"].tolist() == [0, 1, 2] and d["pattern_slot"].tolist() == [3, 3, 3]
    assert np.allclose(d["snr_db"], 12.5)
    with pytest.raises(KeyError):
        E.make_descs("XYZ", 1.0, 1.0, 0, n=1)
    g = E.Geometry()
    assert (g.num_sc, g.grid) == (599, 8386)
    assert E.Geometry(useful_subcarriers=72).num_sc == 71 and E.Geometry(fft_size=2048, useful_subcarriers=1200).num_sc == 1199


def test_shard_ranges_partition_any_total():
    for total in (0, 1, 7, 10000, 1_000_003):
        for world in (1, 2, 3, 8):
            rngs = [S_.shard_range(total, r, world) for r in range(world)]
            assert rngs[0][0] == 0 and rngs[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(rngs, rngs[1:]))
            sizes = [e - b for b, e in rngs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        S_.shard_range(10, 2, 2)


def test_frame_choices_are_a_function_of_the_global_index():
    idx = np.arange(1000)
    whole = S_.frame_choices(42, idx, (3, 4, 8, 2))
    parts = [S_.frame_choices(42, idx[a:b], (3, 4, 8, 2)) for a, b in ((0, 333), (333, 1000))]
    for k in range(4):
        assert np.array_equal(whole[k], np.concatenate([p[k] for p in parts]))
        assert whole[k].min() == 0 and whole[k].max() == (3, 4, 8, 2)[k] - 1
    assert not np.array_equal(whole[0], S_.frame_choices(43, idx, (3, 4, 8, 2))[0])
    counts = np.bincount(S_.frame_choices(7, np.arange(30000), (3,))[0])
    assert np.all(np.abs(counts - 10000) < 400)                                     # roughly uniform


def test_philox_known_answers_and_native_draw_shapes():
    # Random123 known-answer vectors for Philox4x32-10
    r = PH.philox4x32_10(np.array([0], dtype=np.uint64), 0, 0, 0, 0, 0)
    assert [int(x[0]) for x in r] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    r = PH.philox4x32_10(np.array([0x243F6A88], dtype=np.uint64), 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0)
    assert [int(x[0]) for x in r] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    pil = np.sort(np.random.RandomState(0).permutation(14 * 71)[:99])
    d = PH.native_frame_draws(5, 77, pil, 14, 71, 2, 4, 9)
    assert d["u_pilot"].shape == (99,) and d["u_data"].shape == (14 * 71 - 99,)
    assert d["u_angle"].shape == (9, 2, 4, 20) and d["n_re"].shape == (14, 4, 71)
    assert 0 <= d["u_pilot"].min() and d["u_data"].max() < 1
    z = np.concatenate([d["n_re"].ravel(), d["n_im"].ravel()])
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1) < 0.05
    d2 = PH.native_frame_draws(5, 78, pil, 14, 71, 2, 4, 9)
    assert not np.array_equal(d["n_re"], d2["n_re"])


def test_prepare_ml_inputs_matches_reference_semantics():
    dg = importlib.import_module(PKG + ".dataset_generator")
    g = load_golden("small_2x2_eva_d10")
    sample = {"rx_symbols": g["rx"], "H_ls": np.repeat(g["H_ls"][:, :, None, :], 2, 2), "H_true": g["H"],
              "pilot_mask": np.zeros((14, 71), bool)}
    x, y = dg.prepare_ml_inputs(sample, normalize=False)
    assert x.shape == (14, 71, 5) and y.shape == (14, 71, 2)
    assert np.array_equal(x[..., 0], g["rx"][:, 0, :].real) and np.array_equal(y[..., 1], g["H"][:, 0, 0, :].imag)
    xn, yn = dg.prepare_ml_inputs(sample, normalize=True)
    assert np.std(xn[..., :4]) == pytest.approx(1.0, rel=1e-6) and np.std(yn) == pytest.approx(1.0, rel=1e-6)


def test_cubic_operator_reproduces_griddata_cubic():
    g = dict(np.load(os.path.join(GOLDEN, "cubic_small_2x2_eva.npz")))
    S, nsc = 14, 71
    ps, pk = np.unravel_index(g["pilot_indices"], (S, nsc))
    A = ip.cubic_operator(ps, pk, S, nsc)
    assert A.shape == (S * nsc, len(ps)) and A.dtype == np.float32
    mask = np.zeros((S, nsc), bool); mask[ps, pk] = True
    h_p = g["rx"][:, 0, :][mask] / (g["pilot_symbols"] + 1e-12)
    got = (A.astype(np.float64) @ h_p).reshape(S, nsc)
    assert np.max(np.abs(got - g["H_cubic"][:, 0, :])) / np.max(np.abs(g["H_cubic"])) < 1e-5
    assert np.max(np.abs(got[ps, pk] - h_p)) < 1e-5 * np.max(np.abs(h_p))        # exact at the pilots
    with pytest.raises(ValueError):
        ip.build_plan(ps, pk, S, nsc, "cubic")                                   # cubic has no 3-point plan


def test_parallel_plan_builder_equals_serial():
    rs = np.random.RandomState(2)
    pils = [np.sort(rs.permutation(14 * 71)[:60 + 7 * i]) for i in range(6)]
    serial = ip.build_plans_parallel(pils, 14, 71, "linear", workers=0)
    pooled = ip.build_plans_parallel(pils, 14, 71, "linear", workers=3)
    for (i0, w0), (i1, w1), p in zip(serial, pooled, pils):
        assert np.array_equal(i0, i1) and np.array_equal(w0, w1)
        ps, pk = np.unravel_index(p, (14, 71))
        ref = ip.build_plan(ps, pk, 14, 71, "linear")
        assert np.array_equal(i0, ref[0]) and np.array_equal(w0, ref[1])


def _locate_exact(ps, pk, simp, S, nsc):
    """NumPy model of plan_locate_kernel: first simplex (in Qhull's order) whose exact integer orientation tests contain
    the resource element; zero-area simplices skipped; returns dense operator rows [G, P]."""
    G, P = S * nsc, len(ps)
    A = np.zeros((G, P))
    gs, gk = np.divmod(np.arange(G), nsc)
    done = np.zeros(G, bool)
    for a, b, c in simp:
        x0, y0, x1, y1, x2, y2 = ps[a] - gs, pk[a] - gk, ps[b] - gs, pk[b] - gk, ps[c] - gs, pk[c] - gk
        n0, n1, n2 = x1 * y2 - y1 * x2, x2 * y0 - y2 * x0, x0 * y1 - y0 * x1
        den = n0 + n1 + n2
        sgn = np.sign(den)
        hit = (~done) & (den != 0) & (n0 * sgn >= 0) & (n1 * sgn >= 0) & (n2 * sgn >= 0)
        for col, num in ((a, n0), (b, n1), (c, n2)):
            A[hit, col] += (num[hit] / den[hit])
        done |= hit
    return A, done


@pytest.mark.parametrize("S,nsc,P,seed", [(14, 71, 99, 0), (14, 71, 9, 1), (14, 599, 838, 2), (14, 599, 83, 3), (3, 40, 5, 6)])
def test_exact_integer_point_location_reproduces_scipy_plan(S, nsc, P, seed):
    """The device builds the 'linear' plan from Qhull's simplices with exact integer tests; as an operator on the pilot
    values it must equal scipy's find_simplex + transform plan (= griddata), including which elements lie outside the hull."""
    rs = np.random.RandomState(seed)
    pil = np.sort(rs.permutation(S * nsc)[:P])
    ps, pk = np.unravel_index(pil, (S, nsc))
    simp = ip.linear_triangulation(ps, pk)
    assert simp.dtype == np.int32 and simp.shape[1] == 3
    A, inside = _locate_exact(ps, pk, simp, S, nsc)
    idx, w = ip.linear_plan(ps, pk, S, nsc)
    B = np.zeros_like(A)
    for c in range(3):
        np.add.at(B, (np.arange(S * nsc), idx[c].astype(int)), w[c].astype(np.float64))
    assert np.array_equal(inside, np.abs(w).sum(0) > 0)
    assert np.max(np.abs(A - B)) < 2e-7                                  # float32 rounding of the stored host weights
    tri = ip.triangulations_parallel([pil, pil], S, nsc, workers=2)
    assert np.array_equal(tri[0], simp) and np.array_equal(tri[1], simp)


def test_triangulation_iterator_equals_parallel_list():
    rs = np.random.RandomState(7)
    pils = [np.sort(rs.permutation(14 * 71)[:50 + 5 * i]) for i in range(9)]
    want = ip.triangulations_parallel(pils, 14, 71, workers=0)
    for w in (0, 3):
        got = list(ip.triangulations_iter(pils, 14, 71, workers=w))
        assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))


def test_bench_algorithmic_bytes_match_the_survey_formula():
    """bench.py's roofline numerator: 8*S*nsc*(T + R + R*T*(1 + n_est)) + S*nsc + 8*P (SURVEY 8d without the 4-B pilot
    index, which belongs to the shared pattern here)."""
    import bench
    assert bench.algorithmic_bytes_per_frame(14, 599, 2, 2, 838, 2) == 1088498
    assert bench.algorithmic_bytes_per_frame(14, 599, 4, 4, 838, 2) == 3772018
    assert bench.algorithmic_bytes_per_frame(14, 599, 1, 1, 838, 2) == 350530
    assert bench.algorithmic_bytes_per_frame(14, 599, 2, 2, 838, 2) + 4 * 838 == 1091850       # SURVEY's figure for config 2
    # config 4 (dataset schema: float32 mask, 32-B metric record, no per-pilot bytes) and config 5 (one estimate tensor, N = 2048)
    assert bench.algorithmic_bytes_per_frame(14, 599, 2, 2, 0, 2, mask_bytes=4, extra=32) == 1106952 + 32      # SURVEY: 1 106 976 with 24 B of meta
    assert bench.algorithmic_bytes_per_frame(14, 1199, 4, 4, 1678, 1) + 4 * 1678 == 5408442                    # SURVEY's figure for config 5
    assert set(bench.WORKLOADS) >= {"cfg2_2x2_eva_ls_mmse", "cfg3_4x4_etu_mmse", "cfg1_siso_epa_ls", "cfg4_dataset_mixed",
                                    "cfg5_4x4_2048_fullcov"}
    assert bench.WORKLOADS["cfg2_2x2_eva_ls_mmse"]["frames"] == 10000 and bench.HBM_PEAK_GBS == 8000.0
    # the harness-defined covariance of config 5 is Hermitian with T on the diagonal (sum of tap powers 1, J0(0) = 1)
    pil = np.sort(np.random.RandomState(0).permutation(14 * 71)[:40])
    C = bench.fullcov_covariance(1024, 72, 72, 2, "ETU", 300.0, pil)
    assert np.allclose(C, C.conj().T) and np.allclose(np.diag(C).real, 2.0)


def test_utils_host_helpers_follow_the_reference():
    """src/utils.py:13-69, 173-180: seeding, dB helpers, AWGN draw order (real block first, then imaginary), layout helpers."""
    import importlib
    U = importlib.import_module("channel-estimation-in-5g-network_amd.utils")
    assert U.db2linear(10) == 10 ** (10 / 10) and U.db2linear(-3.0) == 10 ** (-0.3)
    assert U.linear2db(0.5) == 10 * np.log10(0.5 + 1e-12)
    U.set_seed(7)
    got = U.awgn_noise((3, 5), snr_db=7.0, signal_power=2.0)
    np.random.seed(7)
    std = np.sqrt(2.0 / 10 ** 0.7 / 2)
    re = np.random.randn(3, 5) * std
    im = np.random.randn(3, 5) * std
    assert np.array_equal(got, re + 1j * im)
    z = np.random.randn(4, 3) + 1j * np.random.randn(4, 3)
    r = U.complex_to_real(z)
    assert r.shape == (4, 3, 2) and np.array_equal(r[..., 0], z.real) and np.array_equal(U.real_to_complex(r), z)
    with pytest.raises(NotImplementedError):
        U.qam_modulation(np.zeros(6, dtype=int), 64)          # rejected before any device work, like the reference (:104)


def test_delaunay_cpu_model_of_the_device_triangulator_is_valid():
    """tools/delaunay_model.py is the CPU model csrc/pattern_kernels.hip follows (row strips merged by edge midpoint, pockets,
    rounds of independent Lawson flips with exact integer predicates): on random lattice patterns its result passes its own
    exact Delaunay check, has as many triangles as Qhull's, and the flips stay within a few per pilot."""
    import importlib.util
    from scipy.spatial import Delaunay
    spec = importlib.util.spec_from_file_location("delaunay_model", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "delaunay_model.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    S, nsc = 14, 73
    for seed, P in ((1, 40), (2, 100), (3, 300), (4, 5)):
        idx = np.sort(np.random.RandomState(seed).permutation(S * nsc)[:P])
        px, py = (idx // nsc).astype(np.int64), (idx % nsc).astype(np.int64)
        tri, nbr = M.initial_triangulation(px, py)
        rounds, flips = M.flip_rounds(px, py, tri, nbr)
        assert M.is_delaunay(px, py, tri, nbr)
        assert len(tri) == len(Delaunay(np.column_stack([px, py])).simplices)
        assert rounds < 40 and flips < 4 * P
        # a region-local schedule (four column regions flipped independently, then one global pass from the marked triangles:
        # built on the device in round 3, dropped for its extra rounds) reaches a valid Delaunay triangulation too
        tri2, nbr2, region = M.initial_triangulation(px, py, nsc=nsc)
        local_rounds, global_rounds, f1, f2 = M.flip_rounds_staged(px, py, tri2, nbr2, region)
        assert M.is_delaunay(px, py, tri2, nbr2) and len(tri2) == len(tri)
        assert max(local_rounds) < 40 and global_rounds < 40 and f1 + f2 < 4 * P


def test_covariance_fingerprint_has_no_cancellation_pattern():
    """the W cache's cheap key is a position-weighted checksum of the covariance's BIT PATTERNS: a +d / -d rectangle
    (which cancels in plain row and column sums), a one-ulp edit (which a float64 row sum absorbs) and a swap of two entries all
    change it; an equal copy does not.  Host arrays and (CPU) tensors."""
    import torch
    be = importlib.import_module("channel-estimation-in-5g-network_amd.baseline_estimators")
    est = be.MMSEEstimator(estimate_statistics=False)
    rs = np.random.RandomState(5)
    cov = (rs.randn(96, 96) + 1j * rs.randn(96, 96)).astype(np.complex128)
    base = est._cov_fingerprint(cov)
    assert est._cov_fingerprint(cov.copy()) == base
    rect = cov.copy(); rect[3, 7] += 0.5; rect[3, 40] -= 0.5; rect[60, 7] -= 0.5; rect[60, 40] += 0.5
    ulp = cov.copy(); ulp[10, 11] = complex(np.nextafter(ulp[10, 11].real, np.inf), ulp[10, 11].imag)
    swap = cov.copy(); swap[20, 1], swap[20, 2] = cov[20, 2], cov[20, 1]
    prints = {base, est._cov_fingerprint(rect), est._cov_fingerprint(ulp), est._cov_fingerprint(swap)}
    assert len(prints) == 4
    for dt in (torch.complex128, torch.complex64):
        t = torch.from_numpy(cov).to(dt)
        tb = est._cov_fingerprint(t)
        assert est._cov_fingerprint(t.clone()) == tb
        t2 = t.clone(); t2[3, 7] += 0.5; t2[3, 40] -= 0.5; t2[60, 7] -= 0.5; t2[60, 40] += 0.5
        t3 = t.clone(); t3[20, 1], t3[20, 2] = t[20, 2].clone(), t[20, 1].clone()
        assert len({tb, est._cov_fingerprint(t2), est._cov_fingerprint(t3)}) == 3
