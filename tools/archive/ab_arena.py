#!/usr/bin/env python3
"""Does it matter whether the seven output tensors of a launch are seven allocations or slices of ONE large allocation?
(The footprint sweep shows 80 000-125 000-frame output sets always on the fast write level, 0.69 of 8 TB/s, while 10 000-frame sets
land between 0.61 and 0.69.)  Config-2 shape, 10 000 frames, `sets` independently allocated output sets of each kind in one process."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
E = importlib.import_module("channel-estimation-in-5g-network_amd.engine")


def carve(arena, specs, align):
    out, off = {}, 0
    for name, (shape, dtype) in specs.items():
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        off = (off + align - 1) // align * align
        out[name] = arena[off:off + n].view(dtype).view(shape)
        off += n
    return out


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    sets = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    geom = E.Geometry(); eng = E.Engine(geom, 0); G = geom.grid
    S, T, R, nsc = geom.num_symbols, geom.num_tx, geom.num_rx, geom.num_sc
    slots = eng.build_patterns(0, [G // 10] * 10, seed=2024, pattern_ids=np.arange(10, dtype=np.uint64), method="linear")
    names = ("tx", "rx", "H", "H_ls", "H_mmse", "pilot_symbols", "pilot_mask")
    d = E.make_descs("EVA", 111.1, 10.0, np.asarray(slots, np.int32)[np.arange(F) // 1024], n=F)
    d_dev = torch.from_numpy(d.view(np.uint8).copy()).to(eng.device)
    specs = {"tx": ((F, S, T, nsc), torch.complex64), "rx": ((F, S, R, nsc), torch.complex64), "H": ((F, S, R, T, nsc), torch.complex64),
             "H_ls": ((F, S, R, T, nsc), torch.complex64), "H_mmse": ((F, S, R, T, nsc), torch.complex64),
             "pilot_symbols": ((F, G // 10), torch.complex64), "pilot_mask": ((F, S, nsc), torch.uint8)}
    total = sum(int(np.prod(s)) * torch.empty((), dtype=t).element_size() for s, t in specs.values())

    def timed(out):
        for _ in range(2):
            eng.run_frames(d_dev, outputs=names, out=out, check=False)
        ks = []
        for i in range(6):
            eng.profile_begin(1)
            eng.run_frames(d_dev, seed=1, frame_offset=i * F, outputs=names, out=out, check=False)
            ks.append(eng.profile_read()[1])
        return float(np.median(ks))

    keep = []
    kinds = ["separate allocations", "one arena, 2 MiB-aligned slices", "one arena, 1 GiB-aligned slices"]
    stagger = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else []
    kinds += [f"arena, 1 GiB-aligned + i x {b} B" for b in stagger if b <= (1 << 27)]
    for kind in kinds:
        for k in range(sets):
            if kind.startswith("separate"):
                out = {n: (torch.zeros if n == "pilot_symbols" else torch.empty)(s, dtype=t, device=eng.device) for n, (s, t) in specs.items()}
            else:
                align = (2 << 20) if "2 MiB" in kind else (1 << 30)
                arena = torch.empty(total + 9 * align, dtype=torch.uint8, device=eng.device)
                base = (-arena.data_ptr()) % align                      # slices relative to an ABSOLUTE boundary
                if " + i x " in kind:
                    step = int(kind.split(" x ")[1].split()[0])
                    out, off = {}, base
                    for i, (name, (shape, dtype)) in enumerate(specs.items()):
                        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
                        off = (off + align - 1) // align * align if (off - base) % align else off
                        a = off + i * step
                        out[name] = arena[a:a + n].view(dtype).view(shape)
                        off = (a + n + align - 1 - base) // align * align + base
                else:
                    out = carve(arena[base:], specs, align)
                out["pilot_symbols"].zero_()
                keep.append(arena)
            keep.append(out)
            ms = timed(out)
            print(f"{kind:40s} set {k}: frame kernel {ms:.3f} ms = {F * 1088498 / ms / 1e9 / 8:.3f} of 8 TB/s", flush=True)
            if not kind.startswith("separate"):
                keep.pop(); keep.pop(); del arena, out
                torch.cuda.empty_cache()
    eng.check_status()


if __name__ == "__main__":
    main()
