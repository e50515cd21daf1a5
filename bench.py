#!/usr/bin/env python3
"""Throughput bench of the fused frame pipeline (simulate + LS + MMSE) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A "step" is one pass of the hot path over one batch of synthetic frames on every GPU.  Default workload: BASELINE.json
configs[1] (2x2 MIMO, 1024-FFT, EVA at 60 km/h = 111.1 Hz @ 2 GHz, SNR 10 dB, 10 % pilots, LS + MMSE, 10 000 frames per
GPU).  Other workloads (`--workload`): cfg1 / cfg3 shapes, `cfg4_dataset_mixed` (the dataset loop of
src/dataset_generator.py:112-125: per-frame random channel type / Doppler / SNR / pilot density, LS + MMSE labels) and
`cfg5_4x4_2048_fullcov` (4x4, 2048-FFT, ETU 300 km/h, full-covariance MMSE src/baseline_estimators.py:181-190 on the MFMA
pipe).  Frames are independent, so N GPUs each run their own contiguous frame range (weak scaling, no data-path
collective); one RCCL gather of the per-rank metric sums closes the job (cfg5 additionally broadcasts the prepared filter
W once, before the timed region).  Rank 0 prints ONE JSON line.  Inputs (frame descriptors, pilot patterns, interpolation
plans) are resident in HBM before the timed region; outputs are written to HBM and stay there.

The default invocation (one GPU, config 2) also reports, in the same line and measured after the main timed region,
`per_frame_patterns` -- the same workload with a new pilot pattern drawn, triangulated and planned on the device for EVERY frame
inside every timed step, as the reference does (src/channel_simulator.py:390-391, src/baseline_estimators.py:65-79) -- and `also`:
five steps each of the cfg1 / cfg3 / cfg4 / cfg5 workloads.  `roofline.frac` is the dominant kernel alone, `roofline.step_frac` the
same bytes over the whole step.

`--gpus N` with N > 1 and no launcher (WORLD_SIZE unset) starts the N ranks itself, one child process per GPU, BEFORE
this process touches the GPU; under `python -m torch.distributed.run` the launcher's environment is used as is.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "channel-estimation-in-5g-network_amd"

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense, v_mfma_f32_32x32x16_f16 / _bf16

WORKLOADS = {
    # kind "frames": one chest_frames_run per step, identical parameters for every frame
    "cfg2_2x2_eva_ls_mmse": dict(kind="frames", T=2, R=2, N=1024, cp=72, U=600, ct="EVA", fd=111.1, snr=10.0, dens=0.10,
                                 frames=10000),
    # BASELINE configs[2] as specified (test_phase2_snr_sweep.py:18-64 of the reference at 4x4): a step = the whole sweep, eight
    # SNR points x 6 250 frames = 50 000 frames, one launch per SNR point into the same output buffers
    "cfg3_4x4_etu_mmse": dict(kind="frames", T=4, R=4, N=1024, cp=72, U=600, ct="ETU", fd=370.4, snr=10.0, dens=0.10,
                              frames=6250, snr_sweep=(-5.0, 0.0, 5.0, 10.0, 15.0, 20.0, 25.0, 30.0)),
    "cfg1_siso_epa_ls": dict(kind="frames", T=1, R=1, N=1024, cp=72, U=600, ct="EPA", fd=5.56, snr=10.0, dens=0.10,
                             frames=10000),
    # kind "dataset": per-frame uniform choice of type / Doppler / SNR / density (run_phase3_dataset_generation.py:104-107),
    # frame i keyed by (seed 42, global index i); descriptors are generated on the host inside the timed step
    "cfg4_dataset_mixed": dict(kind="dataset", T=2, R=2, N=1024, cp=72, U=600, frames=10000,
                               types=("EPA", "EVA", "ETU"), dopplers=(10.0, 50.0, 100.0, 200.0),
                               snrs=(-5.0, 0.0, 5.0, 10.0, 15.0, 20.0, 25.0, 30.0), densities=(0.05, 0.10)),
    # kind "fullcov": simulate -> LS at the pilots -> W h_ls on the f32 MFMA pipe -> interpolation (one pattern, W prepared once)
    # (2 304 frames per step: B * R = 9 216 right-hand sides = 72 row tiles x 14 column tiles = 1 008 GEMM workgroups for the chip's
    # 2 x 256 slots -- two nearly full rounds; 2 048 frames gave 896 = 1.75 rounds)
    "cfg5_4x4_2048_fullcov": dict(kind="fullcov", T=4, R=4, N=2048, cp=144, U=1200, ct="ETU", fd=555.6, snr=10.0, dens=0.10,
                                  frames=2304),
}
PATTERN_GROUP = 1024           # frames that share one pilot pattern / interpolation plan (--pattern-group 1: one per frame)


def algorithmic_bytes_per_frame(S, nsc, T, R, P, n_est=2, mask_bytes=1, pilot_bytes=8, extra=0):
    """Compulsory HBM traffic of one frame when every output is written exactly once as complex64:
    tx + rx + H + n_est estimates, the mask, the pilot symbols (8 B each; the 4-B pilot index of SURVEY's figure belongs
    to the shared pattern here and is not counted per frame), plus `extra` bytes of per-frame records."""
    return 8 * S * nsc * (T + R + R * T * (1 + n_est)) + mask_bytes * S * nsc + pilot_bytes * P + extra


# =====================================================================================================
# N ranks from a plain shell
# =====================================================================================================
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """Start `n` copies of this script, one per GPU, with the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  The parent never initialises the GPU; rank 0's stdout is this process's stdout.
    Returns the job's exit code: 0 only if every rank exited 0; the first failure terminates the remaining ranks."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = None if r == 0 else subprocess.DEVNULL            # one JSON line, from rank 0; stderr of every rank is kept
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in alive:
                    procs[q].terminate()                        # exact PIDs we started
        if alive:
            time.sleep(0.05)
    return rc


def _selftest_spawn(world, rank):
    """CPU-only rehearsal of the launch logic (tests/test_bench_spawn.py): gloo group, one all_reduce, one JSON line."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if os.environ.get("BENCH_SELFTEST_FAIL_RANK") == str(rank):
        sys.exit(7)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "sum": float(t.item())}), flush=True)
    dist.destroy_process_group()


# =====================================================================================================
# CPU baseline: the oracle in *faithful* mode on this box's host cores
# =====================================================================================================
def usable_cores():
    """Host cores this process may actually run on: the scheduler affinity, capped by the container's CPU quota (cgroup v2
    cpu.max / v1 cfs_quota) when there is one -- on a box that shows 256 cores but grants a 16-CPU quota, 256 processes compute
    no faster than 16.  Returns (cores, record of what was seen)."""
    seen = {"os_cpu_count": os.cpu_count()}
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    seen["affinity"] = n
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    seen["cgroup_cpu_quota"] = quota
    if quota is not None:
        n = max(1, min(n, int(quota + 0.999)))
    return n, seen


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def cpu_baseline(workload, budget_s=15.0):
    """Time the oracle in *faithful* mode (the reference's algorithmic structure: full-length Jakes synthesis, per-RE
    channel loop, SciPy griddata per antenna pair, dense PxP inverse) on the host cores of this box: ONE PROCESS PER USABLE
    CORE, a bounded number of frames each (about `budget_s` seconds of wall time)."""
    import multiprocessing as mp
    w = WORKLOADS[workload]
    cores, seen = usable_cores()
    t0 = time.time()
    _cpu_frames((workload, 1, 12345))                           # one frame to size the sample
    one = time.time() - t0
    per_proc = max(1, min(32, int(budget_s / max(one, 1e-3))))
    ctx = mp.get_context("spawn")
    t0 = time.time()
    with ctx.Pool(cores) as pool:
        parts = pool.map(_cpu_frames, [(workload, per_proc, 1000 + i) for i in range(cores)], chunksize=1)
    dt = time.time() - t0
    frames = per_proc * cores
    mse_ls = float(np.mean([p[1] for p in parts])); mse_mm = float(np.mean([p[2] for p in parts]))
    # calibration against the reference itself (tools/ref_ratio.py, run in the build container where the reference is importable;
    # BASELINE.md 3): seconds per frame of the reference / of this faithful oracle, same machine, one thread, config 2
    ratio = None
    try:
        cal = json.load(open(os.path.join(ROOT, "profiles", "cpu_ref_ratio.json")))
        ratio = {"ref_ratio": cal["ref_ratio"], "reference_equivalent": frames / dt / cal["ref_ratio"],
                 "ref_ratio_source": "profiles/cpu_ref_ratio.json (tools/ref_ratio.py: reference %.3f s/frame, oracle faithful %.3f s/frame, "
                                     "1 thread, build container, %s)" % (cal["reference_s_per_frame"], cal["oracle_faithful_s_per_frame"], cal["workload"])}
    except Exception:
        pass
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port", **(ratio or {"ref_ratio": None}),
            "cpu_model": cpu_model(), "host": seen, "one_frame_one_core_s": round(one, 3),
            "est_mse": {"ls": mse_ls, "mmse": mse_mm, "frames": frames},        # same statistic as the GPU line's est_mse, other random frames
            "sample": f"{frames} frames of {workload} ({per_proc}/process x {cores} processes = one per usable host core), oracle faithful mode "
                      f"(float64 NumPy/SciPy restatement with the reference's cost structure), {dt:.1f} s wall"}


def fullcov_covariance(N, cp, U, T, ct, fd, pil, S=14):
    """Harness-defined channel covariance at the pilots for config 5 (the reference has no covariance
    generator): R[(s,k),(s',k')] = T * sum_l p_l exp(-2 pi j (kappa_k - kappa_k') d_l / N) * J0(2 pi f_d (t_s - t_s'));
    the factor T reflects that the estimators observe sum_t H_{r,t} (quirk Q5/Q9)."""
    from scipy.special import j0
    fs = N * 15000.0
    half = U // 2
    nsc = 2 * half - 1
    k = np.arange(nsc)
    kap = np.where(k < half, k - half, k - half + 1)
    # tap table after the reference's rounding (src/channel_simulator.py:75-81); duplicates overwrite (quirk Q1)
    prof = {"EPA": ([0, 30, 70, 90, 110, 190, 410], [0.0, -1.0, -2.0, -3.0, -8.0, -17.2, -20.8]),
            "EVA": ([0, 30, 150, 310, 370, 710, 1090, 1730, 2510], [0.0, -1.5, -1.4, -3.6, -0.6, -9.1, -7.0, -12.0, -16.9]),
            "ETU": ([0, 50, 120, 200, 230, 500, 1600, 2300, 5000], [-1.0, -1.0, -1.0, 0.0, 0.0, 0.0, -3.0, -5.0, -7.0])}[ct]
    p = 10.0 ** (np.asarray(prof[1]) / 10.0)
    p = p / p.sum()
    d = np.round(np.asarray(prof[0], dtype=np.float64) * 1e-9 * fs).astype(int)
    taps = {}
    for pl, dl in zip(p, d):
        taps[int(dl)] = pl                                      # later path with the same delay wins
    ps, pk = np.unravel_index(np.asarray(pil), (S, nsc))
    dk = kap[pk][:, None] - kap[pk][None, :]
    Rf = sum(pl * np.exp(-2j * np.pi * dk * dl / N) for dl, pl in taps.items())
    ts = ps * (N + cp) / fs
    return T * Rf * j0(2 * np.pi * fd * (ts[:, None] - ts[None, :]))


def _cpu_frames(args):
    workload, n, seed = args
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(1)
    except Exception:
        limiter = None
    from oracle import cpu_ref as O
    w = WORKLOADS[workload]
    T, R, N, cp, U = w["T"], w["R"], w["N"], w["cp"], w["U"]
    S = 14
    nsc = len(O.used_indices(N, U))
    rs = np.random.RandomState(seed)
    acc_ls = acc_mm = 0.0
    for _ in range(n):
        if w["kind"] == "dataset":                              # the four np.random.choice of src/dataset_generator.py:114-117
            ct = w["types"][rs.randint(len(w["types"]))]; fd = w["dopplers"][rs.randint(len(w["dopplers"]))]
            snr = w["snrs"][rs.randint(len(w["snrs"]))]; dens = w["densities"][rs.randint(len(w["densities"]))]
        else:
            ct, fd, snr, dens = w["ct"], w["fd"], w["snr"], w["dens"]
        L = len(O.PROFILES[ct][0])
        d = O.legacy_frame_draws(rs, S, nsc, T, R, L, dens)
        pos, mask = O.pattern_from_indices(d["pilot_indices"], S, nsc)
        pil = O.unit_symbols(d["u_pilot"])
        grid = O.insert_pilots(mask, O.unit_symbols(d["u_data"]), pil)
        tx = np.repeat(grid[:, None, :], T, axis=1)
        H = O.channel_frequency_response_faithful(ct, fd, N, cp, S, U, 15000.0, d["u_angle"], d["u_phase"])
        rx = O.apply_channel_faithful(tx, H, snr, d["n_re"], d["n_im"])
        rx4 = O.repeat_over_tx(rx, T)
        if w["kind"] == "fullcov":                              # MMSEEstimator(channel_covariance=R, estimate_statistics=False)
            cov = fullcov_covariance(N, cp, U, T, ct, fd, d["pilot_indices"], S)
            mm = O.mmse_estimate(rx4, pil, mask, pos, snr, covariance=cov, estimate_statistics=False)
            ls = mm
        else:
            ls = O.ls_estimate(rx4, pil, mask, pos, "linear")
            mm = O.mmse_estimate(rx4, pil, mask, pos, snr, dense=True)
        acc_ls += O.evaluate_estimator(H, ls)["mse"]
        acc_mm += O.evaluate_estimator(H, mm)["mse"]
    del limiter
    return n, acc_ls / n, acc_mm / n


# =====================================================================================================
# the measured job
# =====================================================================================================
def _pmc_traffic(workload, F, group):
    """HBM bytes per frame-kernel launch from the committed PMC passes (profiles/pmc_traffic.json), only when the
    record was collected on the CURRENT kernel sources (sha of csrc/frame_kernel.hip and csrc/frame_slab_kernel.hip) and
    this workload; else None."""
    import hashlib
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(tfile))
        sha = lambda f: hashlib.sha256(open(os.path.join(ROOT, PKG, "csrc", f), "rb").read()).hexdigest()[:16]
        if rec.get("workload") == workload and rec.get("frames") == F and group == PATTERN_GROUP and \
                rec.get("kernel_source_sha16") == sha("frame_kernel.hip") and rec.get("frame_slab_kernel_sha16") == sha("frame_slab_kernel.hip"):
            return rec.get("hbm_bytes_per_launch"), f"profiles/pmc_traffic.json @ {rec.get('commit', '?')}"
    except Exception:
        pass
    return None, None


def qhull_replan(eng, IP, slots, S, nsc, workers=0):
    """Give the (device-drawn) pilot patterns in `slots` the REFERENCE's triangulation: scipy.spatial.Delaunay = Qhull, what
    griddata('linear') builds (src/baseline_estimators.py:65-79), uploaded as simplices; point location and barycentric weights
    stay on the device (chest_pattern_set_triangulation, the route of every golden test).  Frames through these plans equal
    the reference's LS / MMSE on the same inputs; the device triangulator's own plans are a valid Delaunay triangulation that
    differs inside co-circular lattice polygons (`per_frame_patterns.plan_parity`)."""
    pils = [eng.get_pilots(int(sl)) for sl in slots]
    for sl, pil, tri in zip(slots, pils, IP.triangulations_iter(pils, S, nsc, workers if len(pils) >= 64 else 0)):
        eng.set_pattern(int(sl), pil, tri)


class Job:
    """Process-group context shared by the legs of one bench process."""
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = args.dist_backend
        if args.gpus != self.world:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        if self.backend == "gloo":
            self.local = self.local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(self.local)
        # collectives run for N > 1, and for N = 1 when --rccl-rehearsal asks for them (a one-rank RCCL communicator: proves the
        # library loads and the job's collectives accept these tensors before the first multi-GPU run)
        self.collectives = self.world > 1 or args.rccl_rehearsal
        if self.collectives:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.world == 1:
                os.environ.setdefault("MASTER_PORT", str(_free_port()))
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local))
            else:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
        self.coll_dev = torch.device("cuda", self.local) if self.backend == "nccl" else torch.device("cpu")

    def close(self):
        import torch.distributed as dist
        if self.collectives:
            dist.destroy_process_group()


def measure_leg(job, args, workload, group, steps, warmup, frames=None, patterns_resident=False):
    """One measured leg: `warmup` untimed steps, then `steps` timed steps of `workload` on every rank, barrier + synchronize on
    both sides, MAX over ranks.  Returns the result record (rank 0) or None (other ranks)."""
    import torch
    import torch.distributed as dist
    world, rank, local, coll_dev = job.world, job.rank, job.local, job.coll_dev
    use_coll = job.collectives
    E = importlib.import_module(PKG + ".engine")
    SH = importlib.import_module(PKG + ".sharding")
    IP = importlib.import_module(PKG + ".interp_plan")
    w = WORKLOADS[workload]
    kind = w["kind"]
    T, R, N, cp, U = w["T"], w["R"], w["N"], w["cp"], w["U"]
    F = frames or w["frames"]
    # launches per step: one per SNR point of a sweep workload (F frames each), else one
    sweep = tuple(w.get("snr_sweep", ())) if (kind == "frames" and not args.single_snr) else ()
    sweep = sweep or ((w["snr"],) if "snr" in w else (None,))
    nrep = len(sweep)
    geom = E.Geometry(fft_size=N, cp_length=cp, num_symbols=14, useful_subcarriers=U, num_tx=T, num_rx=R)
    eng = E.Engine(geom, local)
    S, nsc, G = geom.num_symbols, geom.num_sc, geom.grid
    seed = 42 if kind == "dataset" else 2024
    total_steps = warmup + steps
    group = max(1, group)
    workers = args.plan_workers or min(16, os.cpu_count() or 1)

    # ---- inputs resident in HBM before the timed region: patterns, plans (descriptors too, except for the dataset) -------
    # --patterns device (default): pilots drawn (Philox), triangulated (Delaunay) and turned into plans ON THE DEVICE, one
    # launch for all patterns (Engine.build_patterns); --patterns host: legacy np.random shuffle + Qhull on the host cores.
    t_plan = time.perf_counter()
    device_patterns = args.patterns == "device"
    per_step_patterns = False
    if kind == "dataset":
        # one pattern per (density, group of `group` consecutive GLOBAL frames) over the frame range this job will touch;
        # group == 1: a new pattern for EVERY frame, built inside the timed step from (seed, global frame index)
        dens_list = list(w["densities"])
        npil_of_density = np.asarray([int(G * d) for d in dens_list], dtype=np.int32)
        per_step_patterns = device_patterns and group == 1
        pf_method = "linear_compact" if eng.compact_plans_supported else "linear"     # plans that are read once: the compact format
        first_group, last_group = 0, (total_steps * world * F - 1) // group
        keys = [(di, gidx) for di in range(len(dens_list)) for gidx in range(first_group, last_group + 1)] if not per_step_patterns else []
        if per_step_patterns:
            pils, slots = [], []
            eng.build_patterns(0, np.full(F, npil_of_density.max(), np.int32), seed=seed, pattern_ids=np.arange(F, dtype=np.uint64), method=pf_method)   # reserves the slot range
        elif device_patterns:
            ids = np.asarray([gidx * len(dens_list) + di for di, gidx in keys], dtype=np.uint64)
            slots = eng.build_patterns(0, npil_of_density[[di for di, _ in keys]], seed=seed, pattern_ids=ids, method="linear")
            pils = [None] * len(keys)
        else:
            pils = []
            for di, gidx in keys:
                perm = np.arange(G)
                np.random.RandomState((seed * 1000003 + gidx * 101 + di) % (2 ** 32)).shuffle(perm)   # PilotPattern's shuffle
                pils.append(np.sort(perm[: int(G * dens_list[di])]))
    else:
        npat = 1 if kind == "fullcov" else (F + group - 1) // group
        # one pattern per frame: the pattern belongs to the frame (simulate_transmission draws it), so it is built inside
        # the timed step from (seed, global frame index) unless --patterns-resident asks for the plans to be pre-built
        per_step_patterns = device_patterns and group == 1 and kind == "frames" and not patterns_resident
        # a pattern (and plan) per frame is read once, from HBM: the compact plan format; shared patterns stay in L2: the full one
        pf_method = "linear_compact" if (group == 1 and kind == "frames" and eng.compact_plans_supported) else "linear"
        if device_patterns:
            slots = eng.build_patterns(0, np.full(npat, int(G * w["dens"]), np.int32), seed=2024, pattern_ids=np.arange(npat, dtype=np.uint64), method=pf_method)
            pils = [eng.get_pilots(int(slots[0]))] if kind == "fullcov" else [None] * npat
        else:
            pils = []
            for gidx in range(npat):
                perm = np.arange(G)
                np.random.RandomState(2024 + gidx).shuffle(perm)
                pils.append(np.sort(perm[: int(G * w["dens"])]))
    if not device_patterns:
        eng.max_cached_patterns = max(eng.max_cached_patterns, len(pils) + 8)
        if len(pils) >= 4 * workers:        # Qhull Delaunay on the host (~2 ms per pattern per core); point location on the GPU;
            slots = [eng.pattern_slot(pil, "linear", plan=tri)            # uploads overlap the pool's remaining triangulations
                     for pil, tri in zip(pils, IP.triangulations_iter(pils, S, nsc, workers))]
        else:
            workers = 1
            slots = [eng.pattern_slot(pil, "linear") for pil in pils]
    else:
        workers = 0
    # Patterns that are SHARED by frames are built outside the timed region, so they take the reference's own triangulation
    # (Qhull) at no cost to the measurement: the timed frames are then griddata-identical.  A pattern per frame is triangulated
    # by the device kernel inside the timed step (a valid Delaunay triangulation; divergence measured in `plan_parity`).
    qhull = device_patterns and not per_step_patterns and args.triangulation == "qhull" and len(slots) > 0 and \
        not (group == 1 and kind != "fullcov")
    if qhull:
        torch.cuda.synchronize()
        eng.check_status()
        qhull_replan(eng, IP, slots, S, nsc, workers=min(16, os.cpu_count() or 1))
    torch.cuda.synchronize()
    eng.check_status()
    t_plan = time.perf_counter() - t_plan
    P = int(G * max(w["densities"])) if kind == "dataset" else int(G * w["dens"])
    npatterns = F if per_step_patterns else len(slots)

    names = ["tx", "rx", "H", "H_ls", "H_mmse", "pilot_symbols", "pilot_mask"]
    extra = {}
    if kind == "dataset":
        names = ["tx", "rx", "H", "H_ls", "H_mmse", "pilot_mask_f32", "metrics"]       # the stacked record of run_phase3_dataset_generation.py:135-143 (+H_mmse)
        slot_of = np.full((len(dens_list), last_group + 1), -1, dtype=np.int32)
        for (di, gidx), sl in zip(keys, slots):
            slot_of[di, gidx] = sl
        frame_slots = np.arange(F, dtype=np.int32)
        types_code = np.asarray([E.channel_code(c) for c in w["types"]], dtype=np.int32)
        dop = np.asarray(w["dopplers"], np.float32); snrs = np.asarray(w["snrs"], np.float32)

        def descs_for(b0):
            idx = np.arange(b0, b0 + F, dtype=np.int64)
            ci, di, si, pi = SH.frame_choices(seed, idx, (len(types_code), len(dop), len(snrs), len(dens_list)))
            d = np.empty(F, dtype=E.DESC_DTYPE)
            d["channel_type"] = types_code[ci]; d["doppler_hz"] = dop[di]; d["snr_db"] = snrs[si]
            d["pattern_slot"] = frame_slots if per_step_patterns else slot_of[pi, idx // group]
            return d, (npil_of_density[pi], idx.astype(np.uint64))

        def dataset_step(off):
            d, (npil, ids) = descs_for(off)
            if per_step_patterns:           # every frame draws its own pilot pattern: selection + Delaunay + plan, in place
                eng.build_patterns(0, npil, seed=seed, pattern_ids=ids, method=pf_method)
            return d
        d_first = dataset_step(rank * F)
    else:
        if kind == "fullcov":
            names = ["tx", "rx", "H", "pilot_symbols", "pilot_mask", "h_ls_pilots"]
            slot_of_frame = np.full(F, slots[0], dtype=np.int32)
        else:
            slot_of_frame = np.asarray(slots, dtype=np.int32)[np.arange(F) // group]
            if args.with_metrics:
                names.append("metrics")
        npil_frames = np.full(F, int(G * w["dens"]), np.int32)
        d_sweep = [torch.from_numpy(E.make_descs(w["ct"], w["fd"], snr, slot_of_frame, n=F).view(np.uint8).copy()).to(eng.device)
                   for snr in sweep]
        d_first = d_sweep[0]

    # Output buffers are allocated once and reused by every step.  Which allocation they are matters (a stable 20 % between
    # independently allocated sets, Engine.place_outputs): --placement-tries candidates per tensor group are probed with the real
    # kernel here, before the timed region, and the fastest combination is kept; every level seen is in `config.output_placement`.
    out, placement = eng.place_outputs(d_first, names, seed=seed, tries=args.placement_tries)
    torch.cuda.synchronize()

    if kind == "fullcov":
        # W = R (R + sigma^2 I)^-1 once, on rank 0, float64 on the device; broadcast over RCCL before the timed region
        Pn = len(pils[0])
        W = eng.empty_c64(Pn, Pn)
        t_prep = time.perf_counter()
        if rank == 0:
            cov = fullcov_covariance(N, cp, U, T, w["ct"], w["fd"], pils[0], S)
            W.copy_(eng.mmse_full_prepare(cov, w["snr"]))
            torch.cuda.synchronize()
        t_prep = time.perf_counter() - t_prep
        if use_coll:
            if args.dist_backend == "nccl":
                SH.broadcast_from_root(W, 0, force_collective=True)
            else:
                Wc = W.cpu(); SH.broadcast_from_root(Wc, 0, force_collective=True); W.copy_(Wc)
        # The step's three legs -- frame kernel (HBM-write-bound), filter W h on the f32 MFMA pipe, interpolation + T-fold stores
        # (HBM-write-bound) -- run as a software pipeline over `nch` chunks of the batch on three streams: the MFMA-bound filter of
        # chunk c beside the HBM-bound kernels of chunks c - 1 / c + 1 (--pipeline-chunks 1: strictly in sequence on one stream).
        # the filter is packed into the GEMM's operand format once, with the W it belongs to (before the timed region)
        W = eng.mmse_full_pack_filter(W)
        nch = max(1, min(int(args.pipeline_chunks), F))
        bounds = [(F * c // nch, F * (c + 1) // nch) for c in range(nch)]
        scr = [eng.mmse_full_scratch(slots[0], (b1 - b0) * R) for b0, b1 in bounds]
        out["H_mmse"] = eng.empty_c64(F, S, R, T, nsc)
        if nch > 1:
            sF, sG, sI = (torch.cuda.Stream(device=eng.device) for _ in range(3))
            d_chunks = [d_first[b0 * E.DESC_DTYPE.itemsize: b1 * E.DESC_DTYPE.itemsize] for b0, b1 in bounds]
        extra.update(prepare_s=round(t_prep, 3), pipeline_chunks=nch)

    def step(i):
        off = (i * world + rank) * F
        if kind == "dataset":
            eng.run_frames(dataset_step(off), seed=seed, frame_offset=off, outputs=names, out=out, check=False)
        elif kind == "fullcov" and nch == 1:
            eng.run_frames(d_first, seed=seed, frame_offset=off, outputs=names, out=out, check=False)
            eng.mmse_full_apply_pilots(out["h_ls_pilots"], slots[0], W, out=out["H_mmse"], scratch=scr[0], pads_are_zero=True)
        elif kind == "fullcov":
            main = torch.cuda.current_stream()
            sF.wait_stream(main)                                 # (the previous step has finished with every buffer)
            for c, (b0, b1) in enumerate(bounds):
                oc = {k: out[k][b0:b1] for k in names}
                with torch.cuda.stream(sF):
                    eng.run_frames(d_chunks[c], seed=seed, frame_offset=off + b0, outputs=names, out=oc, check=False)
                    e_f = torch.cuda.Event(); e_f.record()
                with torch.cuda.stream(sG):
                    sG.wait_event(e_f)
                    eng.mmse_full_filter_pilots(oc["h_ls_pilots"], slots[0], W, scr[c], pads_are_zero=True)
                    e_g = torch.cuda.Event(); e_g.record()
                with torch.cuda.stream(sI):
                    sI.wait_event(e_g)
                    eng.mmse_full_interp_pilots(b1 - b0, slots[0], scr[c], out["H_mmse"][b0:b1])
            for st_ in (sF, sG, sI):
                main.wait_stream(st_)
        else:
            for j in range(nrep):
                offj = off * nrep + j * F
                if per_step_patterns:
                    eng.build_patterns(0, npil_frames, seed=2024, pattern_ids=np.arange(offj, offj + F, dtype=np.uint64), method=pf_method)
                eng.run_frames(d_sweep[j], seed=seed, frame_offset=offj, outputs=names, out=out, check=False)

    # est-MSE is evaluate_estimator's job, not part of the LS+MMSE step: one extra UNTIMED launch with the fused
    # error-sum output gives the numbers reported under "est_mse" (fullcov: error sums of the stored tensors).
    if kind == "fullcov":
        step(0)
        es = eng.error_sums(out["H"], out["H_mmse"]).double()
        cnt = float(out["H"].numel())
        summ = torch.stack([es[1], es[0], es[0], torch.tensor(cnt, dtype=torch.float64, device=es.device)])
    elif kind == "dataset":
        mse_out = eng.run_frames(d_first, seed=seed, frame_offset=rank * F, outputs=("metrics",))["metrics"].double()
        summ = torch.stack([mse_out[:, 0].sum(), mse_out[:, 1].sum(), mse_out[:, 2].sum(), mse_out[:, 6].sum()])
    else:
        per_snr = []
        for j in range(nrep):
            m = eng.run_frames(d_sweep[j], seed=seed, frame_offset=rank * F * nrep + j * F, outputs=("metrics",))["metrics"].double()
            per_snr.append(torch.stack([m[:, 0].sum(), m[:, 1].sum(), m[:, 2].sum(), m[:, 6].sum()]))
        summ = torch.stack(per_snr).sum(0)
    for i in range(warmup):
        step(i)
    # ---- the collectives of the job, warmed BEFORE t0 -------------------------------------------------------------------------------
    # RCCL sets up its point-to-point connections lazily, on the first collective that needs them: a gather (send / recv) after
    # nothing but barriers (all-reduces) would pay that set-up inside whatever region it runs in.  Every collective the job uses
    # -- the gather of the metric sums, the all-gather of the per-rank times, the MAX-reduction -- runs once here, untimed.
    def gather_sums(v):
        parts = [torch.zeros_like(v) for _ in range(world)] if rank == 0 else None
        dist.gather(v, parts, dst=0)
        return parts

    def all_ranks(vals):
        v = torch.tensor(vals, dtype=torch.float64, device=coll_dev)
        parts = [torch.zeros_like(v) for _ in range(world)]
        dist.all_gather(parts, v)
        return torch.stack(parts).cpu().numpy()

    if use_coll:
        gather_sums(torch.zeros_like(summ.to(coll_dev)))
        all_ranks([0.0] * 5)
        tw = torch.zeros(1, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    eng.profile_begin(steps * (nch if kind == "fullcov" else nrep))   # library-side HIP events around each kernel, on the launch stream
    # ---- the timed region: EXACTLY `steps` steps between barrier + synchronize on both sides -----------------------------------
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()
        step(warmup + i)
        ev[i][1].record()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0                                    # this rank's own time, before it waits for the others
    if use_coll:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.check_status()
    # ---- the job's single collective, timed on its own: per-rank metric sums gathered to rank 0 (RCCL over xGMI) ------------------
    # (folded into a 40-ms timed region it would be a per-step cost the job does not have: it runs ONCE per job, however many steps)
    collective_ms = None
    if use_coll:
        tc = time.perf_counter()
        gathered = gather_sums(summ.to(coll_dev))
        torch.cuda.synchronize()
        collective_ms = (time.perf_counter() - tc) * 1e3
        if rank == 0:
            summ = torch.stack(gathered).sum(0)
        tmax = torch.tensor([dt, collective_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt, collective_ms = float(tmax[0].item()), float(tmax[1].item())
    call_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))      # one step on the stream (all its kernels)
    kern = eng.profile_read_all()                                        # {kernel name: average ms per launch}
    eng.profile_begin(0)
    # every rank's own figures beside the MAX: the write level of a process's output allocations varies by up to 25 %, so
    # with N ranks `value` (the honest MAX) reads the slowest placement -- per_rank tells that apart from a communication loss
    per_rank = None
    mine = [dt_own / steps * 1e3, kern.get("frame_kernel", 0.0), kern.get("taps_kernel", 0.0), call_ms, kern.get("plan_kernels", 0.0)]
    tab = all_ranks(mine) if use_coll else np.asarray([mine])
    if rank == 0:
        def _stat(col):
            v = tab[:, col]
            return {"min": float(v.min()), "median": float(np.median(v)), "max": float(v.max()), "argmax": int(v.argmax()),
                    "all": [float(x) for x in v]}
        per_rank = {"ms_per_step": _stat(0), "frame_kernel_ms": _stat(1), "taps_kernel_ms": _stat(2), "stream_ms_per_step": _stat(3),
                    "note": "ms_per_step: each rank's own wall time over its timed steps before the closing barrier; `value` uses the MAX over "
                            "ranks of the barrier-to-barrier time"}
        if tab[:, 4].max() > 0:
            per_rank["pattern_kernel_ms"] = _stat(4)

    res = None
    if rank == 0:
        taps_ms, kern_ms = kern.get("taps_kernel", 0.0), kern.get("frame_kernel", 0.0)
        cfgd = {"workload": workload, "frames_per_step_per_gpu": F * nrep, "launches_per_step": nrep, "frames_per_launch": F, "mimo": f"{T}x{R}", "fft_size": N, "pilots": P,
                "pattern_group": group, "patterns": npatterns,
                "pattern_source": (("device: Philox pilot selection; triangulation: Qhull (scipy.spatial.Delaunay, the reference's) on the host, "
                                    "simplices uploaded, point location + weights on the device" if qhull else
                                    "device: Philox pilot selection + device Delaunay + plan in one launch" + (", rebuilt inside every timed step" if per_step_patterns else ""))
                                   if device_patterns else "host: legacy shuffle + Qhull"),
                "triangulation": ("qhull" if (qhull or not device_patterns) else "device"),
                "host_plan_build_s": round(t_plan, 2) if not device_patterns else 0.0, "plan_build_s": round(t_plan, 3),
                "host_plan_workers": workers, "output_placement": placement, "rng": "Philox4x32-10 on device",
                "outputs": names, "parallelism": f"frames sharded x{world}", "dist_backend": args.dist_backend if use_coll else None,
                "collectives": (f"executed ({args.dist_backend}, world {world})" if use_coll else "none (single process)")}
        if kind == "dataset":
            Pavg = float(np.mean([int(G * d) for d in w["densities"]]))
            bytes_frame = algorithmic_bytes_per_frame(S, nsc, T, R, 0, 2, mask_bytes=4, extra=4 * 8)
            cfgd.update(channel=list(w["types"]), doppler_hz=list(w["dopplers"]), snr_db=list(w["snrs"]),
                        pilot_density=list(w["densities"]), pilots_mean=Pavg,
                        path="dataset_generator loop body: 4 choices + simulate_transmission + LSEstimator.estimate + MMSEEstimator.estimate per frame",
                        estimators="LS(linear)+MMSE(diag)", host_descriptor_generation="inside the timed step")
        elif kind == "fullcov":
            bytes_frame = algorithmic_bytes_per_frame(S, nsc, T, R, P, 1)
            cfgd.update(channel=w["ct"], doppler_hz=w["fd"], snr_db=w["snr"], pilot_density=w["dens"],
                        path="simulate_transmission + MMSEEstimator(channel_covariance, estimate_statistics=False).estimate per frame",
                        estimators="MMSE(full covariance, W prepared once in float64 on the device)", **extra)
        else:
            bytes_frame = algorithmic_bytes_per_frame(S, nsc, T, R, P, 2)
            cfgd.update(channel=w["ct"], doppler_hz=w["fd"], snr_db=(list(sweep) if nrep > 1 else w["snr"]), pilot_density=w["dens"],
                        path="simulate_transmission + LSEstimator.estimate + MMSEEstimator.estimate per frame",
                        estimators="LS(linear)+MMSE(diag)")
        if group == 1 and kind != "fullcov":   # a plan of its own per frame has to be read: 16-B record per RE + 4-B pilot index (SURVEY 8d row 4;
            bytes_frame += 16 * G + 4 * P       # the compact format the engine reads instead is smaller -- the algorithmic figure stays SURVEY's)
            cfgd["plan_format"] = pf_method
        if "plan_kernels" in kern:
            cfgd["plan_build_kernel_ms"] = kern["plan_kernels"]
        if kind == "fullcov":
            gemm_ms = kern.get("cgemm_w_kernel", 0.0)                # per launch; a step launches it once per pipeline chunk
            step_ms = dt / steps * 1e3
            hbm = bytes_frame * F / (step_ms * 1e-3) / 1e9
            # The step's dominant kernel is the frame kernel (HBM-bound): `achieved` / `frac` = the bytes IT writes (tx, rx, H, mask,
            # pilot symbols, h_ls_pilots) over its own duration; `step_frac` = SURVEY 8d's bytes of the whole path over the whole step.
            fk_bytes = 8 * S * nsc * (T + R + R * T) + S * nsc + 8 * P + 8 * R * ((P + 15) // 16 * 16)
            fk_ach = fk_bytes * F / nch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
            # The filter W h runs on the fp16 MFMA pipe, each f32 operand split in two fp16 terms (f32-accurate: DESIGN 5 K4): per
            # right-hand side the 3-multiplication complex product is 3 real products x 3 fp16 MFMA terms x 2 P^2 flop = 18 P^2 ISSUED
            # flop for the 8 P^2 the result represents (R right-hand sides per frame, not R T: quirk Q9).
            issued = 18.0 * P * P * R * F / nch / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
            algo = 8.0 * P * P * R * F / nch / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
            roof = {"bound": "hbm", "achieved": fk_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fk_ach / HBM_PEAK_GBS,
                    "step_frac": hbm / HBM_PEAK_GBS, "traffic": None, "kernel": "frame_slab_kernel", "kernel_ms": kern_ms,
                    "frame_kernel_bytes_per_frame": fk_bytes, "kernels_ms": kern, "call_ms": call_ms,
                    "hbm_whole_step": {"bytes_per_frame": bytes_frame, "achieved_GBps": hbm, "frac": hbm / HBM_PEAK_GBS},
                    "mfma": {"kernel": "cgemm_w_kernel_h2", "kernel_ms": gemm_ms, "dtype": "f16 x f16 -> f32 (two-term fp16 split of the f32 operands)",
                             "issued_TFLOPs": issued, "peak": MFMA_F16_PEAK_TFLOPS, "issued_frac": issued / MFMA_F16_PEAK_TFLOPS,
                             "algorithmic_TFLOPs": algo, "algorithmic_vs_f32_mfma_peak": algo / MFMA_F32_PEAK_TFLOPS,
                             "flop_note": "issued = 18 P^2 R per frame on the fp16 pipe; algorithmic = 8 P^2 R per frame (the f32-MFMA kernel of "
                                          "rounds 1-3, chest_tune gemm_f32, issued 6 P^2 R at 0.76 of the 157.3 TFLOP/s f32 peak)"}}
        else:
            achieved = bytes_frame * F / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
            traffic, tsrc = _pmc_traffic(workload, F, group)
            # frac: the dominant kernel alone (algorithmic bytes / its own average duration); step_frac: the same bytes over the
            # whole step as timed (every kernel of the step, launch gaps and -- with per-frame patterns -- the pattern kernel)
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "step_frac": bytes_frame * F * nrep / (dt / steps) / 1e9 / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": tsrc, "kernel": "frame_slab_kernel",
                    "kernel_ms": kern_ms, "taps_kernel_ms": taps_ms, "call_ms": call_ms, "bytes_per_frame": bytes_frame}
        s = summ.cpu().numpy()
        res = {
            "metric": "OFDM frames/sec (LS+MMSE)", "value": F * nrep * world * steps / dt, "unit": "frames/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": cfgd, "roofline": roof, "per_rank": per_rank,
        }
        if collective_ms is not None:
            # the end-of-job gather (once per job, not per step), MAX over ranks of the host-observed time incl. the synchronize after it
            res["collective_ms"] = collective_ms
            res["collective"] = {"op": "gather of the per-rank metric sums to rank 0 (4 float64 per rank)", "backend": args.dist_backend,
                                 "ms": collective_ms, "warmed_before_timed_region": True,
                                 "value_with_collective": F * nrep * world * steps / (dt + collective_ms * 1e-3)}
        if s[3] > 0:
            res["est_mse"] = {"ls": float(s[1] / s[3]), "mmse": float(s[2] / s[3]),
                              "nmse_db_ls": float(10 * np.log10(s[1] / s[0])), "nmse_db_mmse": float(10 * np.log10(s[2] / s[0]))}
            if kind == "fullcov":
                res["est_mse"].pop("ls"); res["est_mse"].pop("nmse_db_ls")
        if kind == "frames" and nrep > 1:
            # per-SNR frame-averaged MSE of rank 0's frames (evaluate_estimator's 'mse') beside the REAL reference's curve
            # (tests/golden/snr_sweep_ref_4x4.json: the loop of test_phase2_snr_sweep.py:18-64, 50 frames per SNR; other random frames)
            ps = torch.stack(per_snr).cpu().numpy()
            curve = {"snr_db": list(sweep), "frames_per_snr": F, "mse_ls": (ps[:, 1] / ps[:, 3]).tolist(), "mse_mmse": (ps[:, 2] / ps[:, 3]).tolist(),
                     "nmse_db_ls": (10 * np.log10(ps[:, 1] / ps[:, 0])).tolist(), "nmse_db_mmse": (10 * np.log10(ps[:, 2] / ps[:, 0])).tolist()}
            try:
                ref = json.load(open(os.path.join(ROOT, "tests", "golden", "snr_sweep_ref_4x4.json")))
                if ref["mimo"] == f"{T}x{R}" and [float(r["snr_db"]) for r in ref["snr"]] == [float(x) for x in sweep]:
                    curve["reference"] = {"source": "tests/golden/snr_sweep_ref_4x4.json (the reference itself, %d frames per SNR)" % ref["frames_per_snr"],
                                          "mse_ls": [r["mse_ls"] for r in ref["snr"]], "mse_mmse": [r["mse_mmse"] for r in ref["snr"]],
                                          "mse_ls_stderr": [r["mse_ls_std"] / ref["frames_per_snr"] ** 0.5 for r in ref["snr"]],
                                          "mse_mmse_stderr": [r["mse_mmse_std"] / ref["frames_per_snr"] ** 0.5 for r in ref["snr"]]}
            except Exception:
                pass
            res["snr_sweep"] = curve
    # hand the leg's HBM back before the next one allocates
    del out
    eng.close()
    torch.cuda.empty_cache()
    return res


def cubic_leg(job, frames=1024, steps=5, warmup=2):
    """`also.ls_cubic_2x2`: LSEstimator('cubic') -- griddata(method='cubic') = Clough-Tocher, src/baseline_estimators.py:65-79 -- on
    config 2's frames, beside the only LS timing the reference publishes (203 ms per 2x2 frame, PHASE_2_BASELINE_ESTIMATORS.md:268,
    timed at test_phase2_comparison.py:76-78 around `ls_estimator.estimate` alone).  The pattern is shared by the batch, so its dense
    Clough-Tocher operator A [G][P] is built ONCE outside the timed region (Qhull's triangulation from the host, everything after it on
    the device: chest_cubic_operator); timed: chest_estimate_dense = pilot division + the real x complex product A [G x P] . h [P x B R T]
    on v_mfma_f32_32x32x2_f32.  Bound: MFMA (4 G P flop per (frame, rx, tx) grid against the 157.3 TFLOP/s f32-matrix peak; the result
    alone is 8 G bytes per grid, an HBM floor ~20x below the matrix time)."""
    import torch
    E = importlib.import_module(PKG + ".engine")
    IP = importlib.import_module(PKG + ".interp_plan")
    w = WORKLOADS["cfg2_2x2_eva_ls_mmse"]
    T, R = w["T"], w["R"]
    geom = E.Geometry(fft_size=w["N"], cp_length=w["cp"], num_symbols=14, useful_subcarriers=w["U"], num_tx=T, num_rx=R)
    eng = E.Engine(geom, job.local)
    S, nsc, G = geom.num_symbols, geom.num_sc, geom.grid
    P = int(G * w["dens"])
    slot0 = int(eng.build_patterns(0, np.asarray([P], np.int32), seed=2024, pattern_ids=np.asarray([0], np.uint64), method=None)[0])
    pil = eng.get_pilots(slot0)
    ps, pk = np.unravel_index(pil.astype(np.int64), (S, nsc))
    t0 = time.perf_counter()
    simp, nbr = IP.delaunay_with_neighbors(ps, pk)
    t_qhull = time.perf_counter() - t0
    slot = eng.pattern_slot(pil, "linear", plan=simp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    A = eng.cubic_operator_from_triangulation(slot, simp, nbr)
    torch.cuda.synchronize(); t_build = time.perf_counter() - t0
    d = E.make_descs(w["ct"], w["fd"], w["snr"], slot, n=frames)
    sim = eng.run_frames(d, seed=2024, outputs=("rx", "pilot_symbols", "H"))
    rx4 = sim["rx"][:, :, :, None, :].expand(frames, S, R, T, nsc).contiguous()      # the callers' rx_4d (src/dataset_generator.py:63-64)
    xp = sim["pilot_symbols"][:, :P].contiguous()
    for _ in range(warmup):
        out = eng.estimate_dense(rx4, xp, slot, A)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        out = eng.estimate_dense(rx4, xp, slot, A)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    es = eng.error_sums(sim["H"], out).cpu().numpy()
    flop = 4.0 * G * P * R * T * frames
    rec = {"value": frames / dt, "unit": "LS('cubic') estimates of a 2x2 frame per second", "ms_per_step": dt * 1e3, "ms_per_frame": dt * 1e3 / frames,
           "frames_per_step": frames, "pilots": P, "bound": "mfma", "achieved_TFLOPs": flop / dt / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
           "frac": flop / dt / 1e12 / MFMA_F32_PEAK_TFLOPS, "flop_note": "4 G P per (frame, rx, tx) grid: real operator x complex right-hand sides, f32 MFMA",
           "operator_build": {"qhull_host_ms": t_qhull * 1e3, "device_ms": t_build * 1e3, "inside_timed_region": False},
           "timed": "chest_estimate_dense only (what the reference's 203 ms times: LSEstimator.estimate); scratch allocation included",
           "reference_published_ms_per_frame": 203.0, "reference_source": "PHASE_2_BASELINE_ESTIMATORS.md:268,293 (unspecified hardware)",
           "est_mse_ls_cubic": float(es[0] / sim["H"].numel()), "triangulation": "qhull"}
    del sim, rx4, out, A
    eng.close()
    torch.cuda.empty_cache()
    return rec


def dropin_leg(job, frames=40, warmup=5):
    """`also.dropin_single_frame`: the reference-named single-frame API, NumPy in / NumPy out, one call per frame as the reference's
    scripts make them -- simulate_transmission + LSEstimator('cubic').estimate + MMSEEstimator().estimate (the two estimators
    test_phase2_comparison.py:76-99 times: 203 ms / 588 ms per 2x2 frame) + LSEstimator('linear').estimate.  PCIe, host RNG replay, a
    new pilot pattern (Qhull on the host) per frame all included: this is latency of the drop-in, never `value`."""
    cs = importlib.import_module(PKG + ".channel_simulator")
    be = importlib.import_module(PKG + ".baseline_estimators")
    cfg = {"ofdm": {"fft_size": 1024, "cp_length": 72, "num_symbols": 14, "useful_subcarriers": 600, "subcarrier_spacing": 15000.0},
           "mimo": {"num_tx_antennas": 2, "num_rx_antennas": 2}, "channel": {"carrier_freq": 2e9}}
    ls_c, ls_l, mm = be.LSEstimator("cubic"), be.LSEstimator("linear"), be.MMSEEstimator()
    np.random.seed(42)
    t = np.zeros(4)

    def frame():
        t0 = time.perf_counter()
        r = cs.simulate_transmission(cfg, "EVA", 111.1, 10, 0.1)
        t1 = time.perf_counter()
        rx4 = np.repeat(r["rx_symbols"][:, :, None, :], 2, axis=2)
        pp = r["pilot_pattern"]
        pos = pp.get_pilot_positions()
        t1b = time.perf_counter()
        ls_c.estimate(rx4, r["pilot_symbols"], pp.pilot_mask, pos)
        t2 = time.perf_counter()
        mm.estimate(rx4, r["pilot_symbols"], pp.pilot_mask, pos, snr_db=10)
        t3 = time.perf_counter()
        ls_l.estimate(rx4, r["pilot_symbols"], pp.pilot_mask, pos)
        t4 = time.perf_counter()
        return np.asarray([t1 - t0, t2 - t1b, t3 - t2, t4 - t3])

    for _ in range(warmup):
        frame()
    for _ in range(frames):
        t += frame()
    t = t / frames * 1e3
    return {"frames": frames, "ms_simulate_transmission": float(t[0]), "ms_ls_cubic_estimate": float(t[1]), "ms_mmse_estimate": float(t[2]),
            "ms_ls_linear_estimate": float(t[3]),
            "reference_published_ms": {"ls_cubic_estimate": 203.0, "mmse_estimate": 588.0,
                                       "source": "PHASE_2_BASELINE_ESTIMATORS.md:268,293; test_phase2_comparison.py:76-99 (unspecified hardware)"},
            "speedup_vs_published": {"ls_cubic_estimate": 203.0 / float(t[1]), "mmse_estimate": 588.0 / float(t[2])},
            "note": "2x2, EVA 111.1 Hz, 10 dB, 10 % pilots; a NEW pilot pattern per frame (Qhull + plan / Clough-Tocher operator built per call), "
                    "NumPy in / NumPy out (PCIe both ways), legacy np.random replay on the host"}


def _brief(res):
    r = res["roofline"]
    b = {"value": res["value"], "ms_per_step": res["ms_per_step"], "frac": r["frac"], "step_frac": r.get("step_frac"),
         "kernel_ms": r.get("kernel_ms"), "bound": r["bound"], "frames_per_step": res["config"]["frames_per_step_per_gpu"],
         "pattern_group": res["config"]["pattern_group"]}
    if "plan_build_kernel_ms" in res["config"]:
        b["pattern_kernel_ms"] = res["config"]["plan_build_kernel_ms"]
    if "mfma" in r:
        b["mfma"] = {k: r["mfma"][k] for k in ("kernel", "kernel_ms", "issued_TFLOPs", "issued_frac", "algorithmic_TFLOPs")}; b["kernels_ms"] = r["kernels_ms"]
    if "snr_sweep" in res:
        b["snr_sweep"] = res["snr_sweep"]
    b["triangulation"] = res["config"].get("triangulation")
    return b


EXTRA_LEGS = (("cfg1_siso_epa_ls", PATTERN_GROUP), ("cfg3_4x4_etu_mmse", PATTERN_GROUP), ("cfg4_dataset_mixed", 1),
              ("cfg5_4x4_2048_fullcov", PATTERN_GROUP))


class _QuietStdout:
    """The contract is ONE JSON line on stdout, but libraries below us write there too (RCCL prints a five-line version banner
    from C when the first communicator comes up; gloo chats as well).  While the job runs, file descriptor 1 points at
    stderr; `restore()` flushes the C library's buffers first -- or the banner would surface at exit, after our line -- and puts
    stdout back for the result."""
    def __init__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def restore(self):
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)


def run(args):
    import torch
    if args.selftest_spawn:
        return _selftest_spawn(int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")))
    quiet = _QuietStdout()
    job = Job(args)
    res = measure_leg(job, args, args.workload, args.pattern_group, args.steps, args.warmup, frames=args.frames,
                      patterns_resident=args.patterns_resident)
    # The default invocation (what the driver runs: BASELINE config 2, shared patterns, one GPU) also measures, after the main
    # timed region: the same workload with the reference's own semantics -- a NEW pilot pattern per frame, drawn, triangulated
    # and planned on the device inside every timed step (src/channel_simulator.py:390-391, src/baseline_estimators.py:65-79) --
    # as `per_frame_patterns`, and five steps each of the other BASELINE shapes as `also`.  `value` stays the main leg's.
    default_run = (args.workload == "cfg2_2x2_eva_ls_mmse" and args.pattern_group == PATTERN_GROUP and args.frames is None
                   and args.patterns == "device" and not args.no_extra_legs and job.world == 1)
    if default_run:
        # (a failing extra leg must not cost the driver its main line: its error is reported in place of its record)
        def leg(*a, **k):
            try:
                return measure_leg(job, args, *a, **k), None
            except Exception as e:                                       # noqa: BLE001 -- reported, not swallowed
                import traceback
                traceback.print_exc(file=sys.stderr)
                torch.cuda.empty_cache()
                return None, f"{type(e).__name__}: {e}"
        pf, err = leg(args.workload, 1, max(5, min(args.steps, 10)), 2)
        if job.rank == 0:
            if pf is None:
                res["per_frame_patterns"] = {"error": err}
            else:
                r = pf["roofline"]
                res["per_frame_patterns"] = {
                    "value": pf["value"], "ms_per_step": pf["ms_per_step"], "frac": r["step_frac"],
                    "pattern_kernel_ms": pf["config"].get("plan_build_kernel_ms"), "frame_kernel_ms": r["kernel_ms"],
                    "taps_kernel_ms": r["taps_kernel_ms"], "frame_kernel_frac": r["frac"], "bytes_per_frame": r["bytes_per_frame"],
                    "plan_format": pf["config"].get("plan_format"), "steps": pf["steps"], "triangulation": "device",
                    "note": "frac = bytes_per_frame x frames/s / 8 TB/s over the WHOLE step (pattern kernel + taps + frame kernel)"}
                # how far the device triangulator's plans are from griddata's on the same frames (tools/plan_parity.py; untimed)
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import plan_parity
                    res["per_frame_patterns"]["plan_parity"] = plan_parity.measure(P=pf["config"]["pilots"], patterns=64, device=job.local)
                except Exception as e:                                   # noqa: BLE001 -- reported in place of the record
                    res["per_frame_patterns"]["plan_parity"] = {"error": f"{type(e).__name__}: {e}"}
        also = {}
        for name, grp in EXTRA_LEGS:
            rec, err = leg(name, grp, 5, 2)
            if job.rank == 0:
                also[name] = _brief(rec) if rec is not None else {"error": err}
        # the two figures the reference itself publishes for this path: LS ('cubic') and MMSE time per 2x2 frame
        for name, fn in (("ls_cubic_2x2", cubic_leg), ("dropin_single_frame", dropin_leg)):
            try:
                also[name] = fn(job)
            except Exception as e:                                           # noqa: BLE001 -- reported in place of the record
                import traceback
                traceback.print_exc(file=sys.stderr)
                torch.cuda.empty_cache()
                also[name] = {"error": f"{type(e).__name__}: {e}"}
        if job.rank == 0:
            res["also"] = also
    job.close()
    # The CPU path beside the GPU number, in the same run, at EVERY N (north_star): rank 0 times it on the host cores after the
    # job -- outside every timed region, the process group already gone, the other ranks exiting.
    if job.rank == 0 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(args.workload, budget_s=args.cpu_budget_s)
    quiet.restore()
    if job.rank == 0:
        print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2_2x2_eva_ls_mmse", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=None, help="frames per step per GPU (default: the config's)")
    ap.add_argument("--pattern-group", type=int, default=PATTERN_GROUP,
                    help="frames sharing one pilot pattern + interpolation plan; 1 = a new random pattern per frame as the "
                         "reference draws them: pilots, Delaunay triangulation and plan are then built on the device INSIDE every "
                         "timed step (--patterns-resident pre-builds them instead)")
    ap.add_argument("--patterns-resident", action="store_true",
                    help="with --pattern-group 1 on a fixed-parameter workload: build the per-frame patterns once before the timed region")
    ap.add_argument("--patterns", default="device", choices=["device", "host"],
                    help="device: pilot patterns drawn, triangulated and planned on the GPU (default); host: legacy np.random "
                         "shuffle + Qhull on the host cores")
    ap.add_argument("--triangulation", default="qhull", choices=["qhull", "device"],
                    help="patterns SHARED by frames (built before the timed region): qhull = the reference's triangulation from "
                         "scipy.spatial.Delaunay, uploaded as simplices (default: the timed frames equal griddata's); device = the "
                         "device triangulator's.  A pattern per frame (--pattern-group 1) is always triangulated on the device")
    ap.add_argument("--plan-workers", type=int, default=0, help="host processes for plan building with --patterns host (0 = min(16, cores))")
    ap.add_argument("--placement-tries", type=int, default=3,
                    help="candidate allocations per output tensor group, probed with the real kernel BEFORE the timed region (clocks "
                         "settled first, two round-robin passes); the fastest combination is kept (Engine.place_outputs: the frame "
                         "kernel's write level is a stable property of the allocation it writes into, profiles/r05_placement_levels.txt). "
                         "1 = take the first allocation")
    ap.add_argument("--pipeline-chunks", type=int, default=1,
                    help="cfg5: chunks of the batch pipelined over three streams (frame kernel | MFMA filter | interpolation); 1 = in sequence")
    ap.add_argument("--single-snr", action="store_true", help="cfg3: one launch at SNR 10 dB per step instead of the eight-point sweep")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0, help="wall-time budget of the cpu_baseline sample (one process per usable host core)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="default invocation only: skip the per_frame_patterns leg and the `also` legs of the other workloads")
    ap.add_argument("--rccl-rehearsal", action="store_true",
                    help="with --gpus 1: initialise the process group anyway (world size 1) and run the job's collectives "
                         "(barrier, gather of the metric sums, max-reduction of the time, cfg5's broadcast of W) through it")
    ap.add_argument("--with-metrics", action="store_true", help="also produce the fused error sums inside the timed step")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank) for real runs; gloo only to rehearse the N>1 control flow with "
                         "several ranks sharing one GPU")
    ap.add_argument("--selftest-spawn", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the ranks ourselves, before anything here touches the GPU (torch is not even imported yet)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    run(args)


if __name__ == "__main__":
    main()
