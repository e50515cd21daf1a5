#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the per-frame-pattern step: bash tools/pmc_perframe.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_pf
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --pattern-group 1 > $O/f.json 2> $O/f.err &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --pattern-group 1 > $O/w.json 2> $O/w.err
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for tag, name in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    for f in glob.glob(f"gpurun_out/pmc_pf/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if r["Counter_Name"] == name and int(r["Grid_Size"]) >= 10000 * 256 and ("frame_slab" in k or "pattern_build" in k or "taps_kernel" in k):
                acc[k.split("(")[0][:60]][name].append(float(r["Counter_Value"]))
print("per launch of 10 000 frames / patterns (last three launches averaged); rocprofv3 KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md")
for k, d in acc.items():
    rd = 2.0 * 1024 * sum(d["FETCH_SIZE"][-3:]) / max(1, len(d["FETCH_SIZE"][-3:]))
    wr = 1024 * sum(d["WRITE_SIZE"][-3:]) / max(1, len(d["WRITE_SIZE"][-3:]))
    print(f"{k:62s} read {rd / 1e9:7.3f} GB   written {wr / 1e9:7.3f} GB")
PY
