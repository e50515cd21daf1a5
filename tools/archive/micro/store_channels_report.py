#!/usr/bin/env python3
"""Per-channel (16 L2 channels x 8 XCDs) view of the write counters rocprofv3 collected around tools/micro/store_channels:
   python tools/micro/store_channels_report.py <results.json> <stdout of store_channels>"""
import json
import sys
import numpy as np

d = json.load(open(sys.argv[1]))["rocprofiler-sdk-tool"][0]
names = {c["id"]["handle"]: c["name"] for c in d["counters"]}
lines = [l.split() for l in open(sys.argv[2]) if l.startswith("dispatch")]
info = {int(l[1]): (int(l[5]), int(l[7]), float(l[8])) for l in lines}
mode = {int(l[1]): (l[12] if len(l) > 12 and not l[12].startswith("L@") else "") for l in lines}          # dispatch -> (set, which, ms)
for rec in d["callback_records"]["counter_collection"]:
    di = rec["dispatch_data"]["dispatch_info"]["dispatch_id"] - 1
    dur = (rec["dispatch_data"]["end_timestamp"] - rec["dispatch_data"]["start_timestamp"]) * 1e-6
    per = {}
    for r in rec["records"]:
        per.setdefault(names[r["counter_id"]["handle"]], []).append(r["value"])
    st, which, ms = info.get(di, (-1, -1, 0.0))
    out = [f"disp {di:3d} set {st} {mode.get(di, '')} {which} {dur:6.3f} ms"]
    for n, v in sorted(per.items()):
        a = np.array(v).reshape(8, 16)                                            # [xcc][channel]
        hot = a.sum(0).reshape(2, 8).sum(0)
        out.append(f"{n.replace('TCC_', '')}: hot ch%8 = {int(hot.argmax())} ({hot.max() / max(np.median(hot), 1e-9):.1f}x median) sum {a.sum():.4g} max/mean {a.max() / max(a.mean(), 1e-9):.3f} min/mean {a.min() / max(a.mean(), 1e-9):.3f}"
                   f" ch-sum max/mean {a.sum(0).max() / max(a.sum(0).mean(), 1e-9):.3f}")
    print(" | ".join(out))
    if "-v" in sys.argv:
        for n, v in sorted(per.items()):
            a = np.array(v).reshape(8, 16)
            print("   ", n, "per channel (sum over XCDs):", " ".join(f"{x:.3g}" for x in a.sum(0)))
            print("   ", n, "per XCD (sum over channels):", " ".join(f"{x:.3g}" for x in a.sum(1)))
