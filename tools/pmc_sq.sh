#!/bin/bash
# SQ issue counters of the frame kernel (diagnostic): bash tools/pmc_sq.sh [extra bench.py args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $O/a.json 2> $O/a.err &&
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $O/b -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $O/b.json 2> $O/b.err
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in "ab":
    for f in glob.glob(f"gpurun_out/pmc_sq/{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "frame_slab" in k or "taps" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            print(k)
            for c, v in sorted(d.items()):
                print(f"   {c:24s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
