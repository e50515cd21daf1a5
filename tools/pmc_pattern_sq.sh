#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${ROUND:-r04}/pmc_pat
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -E "Counter_Name.*(ICACHE|IFETCH|SQC_|INST_LEVEL|SQ_INSTS_BRANCH|SQ_WAIT_INST|SQ_INST_CYCLES)" | head -60 > $O/../icache_counters.txt; cat $O/../icache_counters.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/bench_patterns.py 4096 > $O/a.txt 2> $O/a.err
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/b -- python3 $R/tools/bench_patterns.py 4096 > $O/b.txt 2> $O/b.err
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/c -- python3 $R/tools/bench_patterns.py 4096 > $O/c.txt 2> $O/c.err; tail -3 $O/c.err
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in "abc":
    for f in glob.glob("gpurun_out/" + __import__("os").environ.get("ROUND", "r04") + f"/pmc_pat/{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "pattern_build" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(acc.items()):
            print(f"   {c:24s} n={len(v)} first case (P=838 linear) mean={sum(v[:5])/5:.4g}   P=83 linear mean={sum(v[10:15])/5:.4g}")
PY
