#!/usr/bin/env python3
"""Throughput bench of the fused frame pipeline (simulate + LS + MMSE) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic frames: BASELINE.json configs[1]
(2x2 MIMO, 1024-FFT, EVA at 60 km/h = 111.1 Hz @ 2 GHz, SNR 10 dB, 10 % pilots, LS + MMSE, 10 000
frames) per GPU.  Frames are independent, so N GPUs each run their own contiguous frame range
(weak scaling, no data-path collective); one RCCL all_gather of the per-rank metric sums closes the
job.  Rank 0 prints ONE JSON line.  Inputs (frame descriptors, pilot patterns, interpolation plans)
are resident in HBM before the timed region; outputs are written to HBM and stay there.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "channel-estimation-in-5g-network_amd"

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (T, R, fft, cp, U, channel, doppler_hz, snr_db, density, frames_per_step)
    "cfg2_2x2_eva_ls_mmse": (2, 2, 1024, 72, 600, "EVA", 111.1, 10.0, 0.10, 10000),
    "cfg3_4x4_etu_mmse": (4, 4, 1024, 72, 600, "ETU", 370.4, 10.0, 0.10, 6250),
    "cfg1_siso_epa_ls": (1, 1, 1024, 72, 600, "EPA", 5.56, 10.0, 0.10, 10000),
}
PATTERN_GROUP = 1024           # frames that share one pilot pattern / interpolation plan (--pattern-group 1: one per frame)


def algorithmic_bytes_per_frame(S, nsc, T, R, P, n_est=2):
    """Compulsory HBM traffic of one frame when every output is written exactly once as complex64
    (SURVEY 8d): tx + rx + H + n_est estimates, 1-B mask per RE, 8-B pilot symbol per pilot.
    (SURVEY's figure adds a 4-B pilot index per pilot; here the index list belongs to the shared
    pattern, so it is not counted per frame.)"""
    return 8 * S * nsc * (T + R + R * T * (1 + n_est)) + S * nsc + 8 * P


def cpu_baseline(workload, budget_s=20.0):
    """Time the oracle in *faithful* mode (the reference's algorithmic structure: full-length Jakes
    synthesis, per-RE channel loop, SciPy griddata per antenna pair, dense PxP inverse) on the host
    cores of this box.  One process per core, a bounded number of frames each."""
    import multiprocessing as mp
    T, R, N, cp, U, ct, fd, snr, dens, _ = WORKLOADS[workload]
    cores = min(os.cpu_count() or 1, 16)
    # one frame costs ~0.5-1 s per core in faithful mode; scale the sample to ~budget_s of wall time
    t0 = time.time()
    _cpu_frames((workload, 1, 12345))                         # one frame to size the sample
    one = time.time() - t0
    per_proc = max(1, int(budget_s / max(one, 1e-3)))
    per_proc = min(per_proc, 32)
    ctx = mp.get_context("spawn")
    t0 = time.time()
    with ctx.Pool(cores) as pool:
        parts = pool.map(_cpu_frames, [(workload, per_proc, 1000 + i) for i in range(cores)])
    dt = time.time() - t0
    frames = per_proc * cores
    mse_ls = float(np.mean([p[1] for p in parts])); mse_mm = float(np.mean([p[2] for p in parts]))
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "est_mse": {"ls": mse_ls, "mmse": mse_mm, "frames": frames},        # same statistic as the GPU line's est_mse, other random frames
            "sample": f"{frames} frames of {workload} ({per_proc}/process x {cores} processes), oracle faithful mode "
                      f"(float64 NumPy/SciPy restatement with the reference's cost structure), {dt:.1f} s wall"}


def _cpu_frames(args):
    workload, n, seed = args
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(1)
    except Exception:
        limiter = None
    from oracle import cpu_ref as O
    T, R, N, cp, U, ct, fd, snr, dens, _ = WORKLOADS[workload]
    S = 14
    nsc = len(O.used_indices(N, U))
    L = len(O.PROFILES[ct][0])
    rs = np.random.RandomState(seed)
    acc_ls = acc_mm = 0.0
    for _ in range(n):
        d = O.legacy_frame_draws(rs, S, nsc, T, R, L, dens)
        pos, mask = O.pattern_from_indices(d["pilot_indices"], S, nsc)
        pil = O.unit_symbols(d["u_pilot"])
        grid = O.insert_pilots(mask, O.unit_symbols(d["u_data"]), pil)
        tx = np.repeat(grid[:, None, :], T, axis=1)
        H = O.channel_frequency_response_faithful(ct, fd, N, cp, S, U, 15000.0, d["u_angle"], d["u_phase"])
        rx = O.apply_channel_faithful(tx, H, snr, d["n_re"], d["n_im"])
        rx4 = O.repeat_over_tx(rx, T)
        ls = O.ls_estimate(rx4, pil, mask, pos, "linear")
        mm = O.mmse_estimate(rx4, pil, mask, pos, snr, dense=True)
        acc_ls += O.evaluate_estimator(H, ls)["mse"]
        acc_mm += O.evaluate_estimator(H, mm)["mse"]
    del limiter
    return n, acc_ls / n, acc_mm / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2_2x2_eva_ls_mmse", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=None, help="frames per step per GPU (default: the config's)")
    ap.add_argument("--pattern-group", type=int, default=PATTERN_GROUP,
                    help="frames sharing one pilot pattern + interpolation plan; 1 = a new random pattern per frame as the "
                         "reference draws them (Qhull triangulation on the host, plan on the device, BEFORE the timed region)")
    ap.add_argument("--plan-workers", type=int, default=0, help="host processes for plan building (0 = min(16, cores))")
    ap.add_argument("--placement-tries", type=int, default=4,
                    help="candidate allocations per output tensor group; the fastest placement is kept (1 = take the first)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-metrics", action="store_true", help="also produce the fused error sums inside the timed step")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank) for real runs; gloo only to rehearse the N>1 control flow with "
                         "several ranks sharing one GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dist_backend == "gloo":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    coll_dev = torch.device("cuda", local) if args.dist_backend == "nccl" else torch.device("cpu")

    E = importlib.import_module(PKG + ".engine")
    T, R, N, cp, U, ct, fd, snr, dens, F = WORKLOADS[args.workload]
    if args.frames:
        F = args.frames
    geom = E.Geometry(fft_size=N, cp_length=cp, num_symbols=14, useful_subcarriers=U, num_tx=T, num_rx=R)
    eng = E.Engine(geom, local)
    S, nsc, G = geom.num_symbols, geom.num_sc, geom.grid
    P = int(G * dens)

    # ---- inputs resident in HBM before the timed region: patterns, plans, descriptors -------------
    group = max(1, args.pattern_group)
    npat = (F + group - 1) // group
    pils = []
    for gidx in range(npat):
        rs = np.random.RandomState(2024 + gidx)                # legacy shuffle, as PilotPattern does
        perm = np.arange(G)
        rs.shuffle(perm)
        pils.append(np.sort(perm[:P]))
    t_plan = time.perf_counter()
    workers = args.plan_workers or min(16, os.cpu_count() or 1)
    eng.max_cached_patterns = max(eng.max_cached_patterns, npat + 8)
    if npat >= 4 * workers:                 # Qhull Delaunay on the host (~2 ms per pattern per core); point location on the GPU.
        IP = importlib.import_module(PKG + ".interp_plan")           # Uploads overlap the pool's remaining triangulations.
        slots = [eng.pattern_slot(pil, "linear", plan=tri) for pil, tri in zip(pils, IP.triangulations_iter(pils, S, nsc, workers))]
    else:
        workers = 1
        slots = [eng.pattern_slot(pil, "linear") for pil in pils]
    t_plan = time.perf_counter() - t_plan
    slot_of_frame = np.asarray(slots, dtype=np.int32)[np.arange(F) // group]
    descs = E.make_descs(ct, fd, snr, slot_of_frame, n=F)
    d_dev = torch.from_numpy(descs.view(np.uint8).copy()).to(eng.device)
    names = ["tx", "rx", "H", "H_ls", "H_mmse", "pilot_symbols", "pilot_mask"]
    if args.with_metrics:
        names.append("metrics")
    # Output buffers are allocated once and reused by every step.  Where they live matters on MI355X (Engine.place_outputs,
    # DESIGN.md section 5): probe a few placements per tensor group with the real kernel and keep the fastest.
    out, placement = eng.place_outputs(d_dev, names, seed=2024, tries=args.placement_tries)
    torch.cuda.synchronize()

    def step(i):
        off = (i * world + rank) * F
        eng.run_frames(d_dev, seed=2024, frame_offset=off, outputs=names, out=out, check=False)

    # est-MSE is evaluate_estimator's job, not part of the LS+MMSE step: one extra UNTIMED launch with the fused
    # error-sum output gives the numbers reported under "est_mse".
    mse_out = eng.run_frames(d_dev, seed=2024, frame_offset=rank * F, outputs=("metrics",))["metrics"].double()
    summ = torch.stack([mse_out[:, 0].sum(), mse_out[:, 1].sum(), mse_out[:, 2].sum(), mse_out[:, 6].sum()])   # this rank's partial sums
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    eng.profile_begin(args.steps)            # library-side HIP events around each kernel, on the launch stream
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        step(args.warmup + i)
        ev[i][1].record()
    # the job's single collective: per-rank metric sums to every rank (RCCL over xGMI)
    if world > 1:
        summ_c = summ.to(coll_dev)
        gathered = [torch.zeros_like(summ_c) for _ in range(world)]
        dist.all_gather(gathered, summ_c)
        summ = torch.stack(gathered).sum(0)
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.check_status()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    call_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))      # one chest_frames_run call = taps + frame kernels
    taps_ms, kern_ms, _ = eng.profile_read()                             # each kernel on its own
    eng.profile_begin(0)

    if rank == 0:
        bytes_frame = algorithmic_bytes_per_frame(S, nsc, T, R, P, 2)
        if group == 1:          # a plan of its own per frame has to be read: 16-B record per RE + 4-B pilot index
            bytes_frame += 16 * G + 4 * P
        achieved = bytes_frame * F / (kern_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                rec = json.load(open(tfile))
                if rec.get("workload") == args.workload and rec.get("frames") == F and group == PATTERN_GROUP:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        s = summ.cpu().numpy()
        res = {
            "metric": "OFDM frames/sec (LS+MMSE)", "value": F * world * args.steps / dt, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "frames_per_step_per_gpu": F, "mimo": f"{T}x{R}", "fft_size": N,
                       "channel": ct, "doppler_hz": fd, "snr_db": snr, "pilot_density": dens, "pilots": P,
                       "path": "simulate_transmission + LSEstimator.estimate + MMSEEstimator.estimate per frame",
                       "estimators": "LS(linear)+MMSE(diag)", "pattern_group": group, "patterns": npat,
                       "host_plan_build_s": round(t_plan, 2), "host_plan_workers": workers,
                       "output_placement": placement,
                       "rng": "Philox4x32-10 on device", "outputs": names, "parallelism": f"frames sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "frame_kernel",
                         "kernel_ms": kern_ms, "taps_kernel_ms": taps_ms, "call_ms": call_ms,
                         "bytes_per_frame": bytes_frame},
        }
        if s[3] > 0:
            res["est_mse"] = {"ls": float(s[1] / s[3]), "mmse": float(s[2] / s[3]),
                              "nmse_db_ls": float(10 * np.log10(s[1] / s[0])), "nmse_db_mmse": float(10 * np.log10(s[2] / s[0]))}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
