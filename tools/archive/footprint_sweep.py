#!/usr/bin/env python3
"""Per-frame time of the full pipeline vs batch size (output footprint), plus the PCIe-inclusive rate.  GPU only."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
E = importlib.import_module("channel-estimation-in-5g-network_amd.engine")


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1000,2500,5000,10000,20000,40000,80000,125000", help="frames per launch (one resident output set each)")
    ap.add_argument("--sets", type=int, default=2, help="independently allocated output sets per size (placement spread)")
    ap.add_argument("--no-pcie", action="store_true")
    args = ap.parse_args()
    sizes = [int(x) for x in args.sizes.split(",")]
    geom = E.Geometry(); eng = E.Engine(geom, 0); G = geom.grid
    npat = (max(sizes) + 1023) // 1024
    slots = eng.build_patterns(0, [G // 10] * npat, seed=2024, pattern_ids=np.arange(npat, dtype=np.uint64), method="linear")
    names = ("tx", "rx", "H", "H_ls", "H_mmse", "pilot_symbols", "pilot_mask")
    print("# BASELINE config 2 shape (2x2, 1024-FFT, EVA, LS + MMSE, all seven tensors, one pattern per 1 024 frames); every size writes ONE\n"
          "# resident output set of that many frames (no buffer reuse inside a launch); kernel = the library's HIP events around frame_slab_kernel")
    for F in sizes:
        d = E.make_descs("EVA", 111.1, 10.0, np.asarray(slots, np.int32)[np.arange(F) // 1024], n=F)
        d_dev = torch.from_numpy(d.view(np.uint8).copy()).to(eng.device)
        keep = []
        for k in range(args.sets):
            try:
                out = eng.run_frames(d_dev, outputs=names)
            except torch.cuda.OutOfMemoryError:
                print(f"F={F:6d}  set {k}: out of device memory", flush=True)
                break
            keep.append(out)                                     # (kept alive so that the next set gets other memory)
            for _ in range(2):
                eng.run_frames(d_dev, outputs=names, out=out, check=False)
            torch.cuda.synchronize()
            ts, ks = [], []
            for i in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                eng.profile_begin(1)
                a.record(); eng.run_frames(d_dev, seed=1, frame_offset=i * F, outputs=names, out=out, check=False); b.record()
                torch.cuda.synchronize(); ts.append(a.elapsed_time(b)); ks.append(eng.profile_read()[1])
            med, kmed = float(np.median(ts)), float(np.median(ks))
            print(f"F={F:6d}  set {k}  footprint {F * 1.0885e-3:7.2f} GB   step {med:8.3f} ms   kernel {kmed:8.3f} ms = {F * 1088498 / kmed / 1e9:6.3f} TB/s"
                  f" = {F * 1088498 / kmed / 1e9 / 8:5.3f} of 8 TB/s   {F / med / 1e3:6.3f} M frames/s", flush=True)
        eng.profile_begin(0)
        del keep, out
        torch.cuda.empty_cache()
    if args.no_pcie:
        return
    slots = [int(s) for s in slots[:2]]
    # PCIe-inclusive: device generation + copy of every output tensor to pageable host NumPy (what the drop-in API does)
    F = 2000
    d = E.make_descs("EVA", 111.1, 10.0, np.asarray(slots, np.int32)[np.arange(F) // 1024], n=F)
    t0 = time.time()
    for i in range(3):
        out = eng.run_frames(d, seed=2, frame_offset=i * F, outputs=names)
        host = {k: v.cpu().numpy() for k, v in out.items()}
    dt = (time.time() - t0) / 3
    print(f"PCIe-inclusive (to pageable NumPy): {F / dt:9.0f} frames/s  ({F * 1.0885e-3 / dt:5.2f} GB/s)", flush=True)
    pin = {k: torch.empty(v.shape, dtype=v.dtype).pin_memory() for k, v in out.items()}
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(3):
        out = eng.run_frames(d, seed=2, frame_offset=i * F, outputs=names, check=False)
        for k in out: pin[k].copy_(out[k], non_blocking=True)
        torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    print(f"PCIe-inclusive (to pinned host buffers):  {F / dt:9.0f} frames/s  ({F * 1.0885e-3 / dt:5.2f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
