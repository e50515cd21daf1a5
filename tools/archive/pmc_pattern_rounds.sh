#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${ROUND:-r04}/pmc_rounds
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/pattern_rounds.py ${1:-838} > $O/a.txt 2> $O/a.err
cd $R
python3 - <<'PY'
import csv, glob, collections
caps = (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 40)
for f in glob.glob("gpurun_out/" + __import__("os").environ.get("ROUND", "r04") + "/pmc_rounds/a/**/*counter_collection.csv", recursive=True):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pattern_build" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
    v = [x / 4096 for x in rows["SQ_INSTS_VALU"][1:]]            # [before flips, cap 1, cap 2, ...]
    base = v[0]
    print(f"vector instructions per pattern before the flips: {base:,.0f}")
    prev, pc = base, 0
    for cap, x in zip(caps, v[1:]):
        print(f"   rounds <= {cap:2d}: flips so far {x - base:8,.0f}   (+{(x - prev) / (cap - pc):7,.0f} per round)")
        prev, pc = x, cap
PY
