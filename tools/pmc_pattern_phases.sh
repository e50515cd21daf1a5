#!/bin/bash
# instructions per phase of pattern_build_kernel<compact>: SQ instruction counters of the kernel truncated after phase k
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${ROUND:-r04}/pmc_phases
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/pattern_phases.py ${1:-838} > $O/a.txt 2> $O/a.err
cd $R
python3 - <<'PY'
import csv, glob, collections
names = ["selection", "slot table", "row tables + strips", "pockets", "flips", "compact table + walk"]
for f in glob.glob("gpurun_out/" + __import__("os").environ.get("ROUND", "r04") + "/pmc_phases/a/**/*counter_collection.csv", recursive=True):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pattern_build" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
    n = 4096
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "GRBM_GUI_ACTIVE"):
        v = rows[c][1:]                                  # first launch = the untruncated warm-up
        per_k = [sum(v[3 * i:3 * i + 3]) / 3 for i in range(6)]      # k = 1..5, 0
        cum = per_k[:5] + [per_k[5]]
        diffs = [cum[0]] + [cum[i] - cum[i - 1] for i in range(1, 6)]
        unit = 1 / n if c != "GRBM_GUI_ACTIVE" else 1 / 8 / n * 256
        print(f"{c} per pattern (GRBM: CU-cycles per pattern): total {cum[5] * unit:,.0f}")
        for nm, d in zip(names, diffs):
            print(f"    {nm:24s} {d * unit:12,.0f}  {d / cum[5]:6.1%}")
PY
