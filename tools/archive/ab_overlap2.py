#!/usr/bin/env python3
"""Steps with a new pilot pattern per frame as a software pipeline ACROSS steps: the pattern build of step i + 1 (vector-ALU-bound, no HBM
traffic) on one stream beside the frame kernel of step i (HBM-write-bound) on another, two pattern pools.  With three frame workgroups per
CU (156 KB of LDS) no pattern workgroup fits beside them; `frame_min_waves = 4` selects the 128-VGPR frame kernel (two per CU, 104 KB), which
leaves room for one pattern workgroup (31.7 KB).  Compared with the same steps in plain stream order.  GPU only."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
E = importlib.import_module("channel-estimation-in-5g-network_amd.engine")


def run(min_waves, overlap, F=10000, steps=12, warm=3):
    geom = E.Geometry()
    eng = E.Engine(geom, 0)
    eng.tune("frame_min_waves", min_waves)
    G = geom.grid
    P = G // 10
    npil = np.full(F, P, np.int32)
    names = ["tx", "rx", "H", "H_ls", "H_mmse", "pilot_symbols", "pilot_mask"]
    descs = []
    for pool in (0, 1):
        eng.build_patterns(pool * F, npil, seed=2024, pattern_ids=np.arange(F, dtype=np.uint64), method="linear_compact")
        d = E.make_descs("EVA", 111.1, 10.0, np.arange(pool * F, (pool + 1) * F, dtype=np.int32), n=F)
        descs.append(torch.from_numpy(d.view(np.uint8).copy()).to(eng.device))
    out = eng.run_frames(descs[0], seed=1, outputs=names)
    torch.cuda.synchronize()
    sA, sB = torch.cuda.Stream(device=eng.device), torch.cuda.Stream(device=eng.device)
    main = torch.cuda.current_stream()

    def patterns(i):
        eng.build_patterns((i % 2) * F, npil, seed=2024, pattern_ids=np.arange(i * F, (i + 1) * F, dtype=np.uint64), method="linear_compact")

    def frames(i):
        eng.run_frames(descs[i % 2], seed=1, frame_offset=i * F, outputs=names, out=out, check=False)

    def steps_serial(i0, n):
        for i in range(i0, i0 + n):
            patterns(i); frames(i)

    def steps_pipelined(i0, n):
        sA.wait_stream(main); sB.wait_stream(main)
        eP, eF = {}, {}
        with torch.cuda.stream(sB):
            patterns(i0); eP[i0] = torch.cuda.Event(); eP[i0].record()
        for i in range(i0, i0 + n):
            with torch.cuda.stream(sA):
                sA.wait_event(eP[i]); frames(i); eF[i] = torch.cuda.Event(); eF[i].record()
            if i + 1 < i0 + n:
                with torch.cuda.stream(sB):
                    if i - 1 in eF: sB.wait_event(eF[i - 1])           # the pool about to be rebuilt was last read by step i - 1
                    patterns(i + 1); eP[i + 1] = torch.cuda.Event(); eP[i + 1].record()
        main.wait_stream(sA); main.wait_stream(sB)

    fn = steps_pipelined if overlap else steps_serial
    fn(0, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(warm, steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    eng.check_status()
    eng.close()
    return dt


if __name__ == "__main__":
    for mw in (0, 4):
        a, b = run(mw, False), run(mw, True)
        print(f"frame kernel {'3 workgroups per CU (96 VGPRs)' if mw == 0 else '2 workgroups per CU (128 VGPRs)'}: stream order {a:.3f} ms per step = {1226026 * 1e4 / a / 1e9 / 8:.3f} of 8 TB/s;"
              f"  pattern build of step i + 1 beside the frame kernel of step i {b:.3f} ms = {1226026 * 1e4 / b / 1e9 / 8:.3f}", flush=True)
