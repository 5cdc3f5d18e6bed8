#!/bin/bash
# one PMC pass: VALU/SALU/LDS instruction counts and wave cycles of the fused kernel
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/qpmc_${1:-x}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p0 -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/p0.log 2>&1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p0/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "fused_block_kernel" not in row.get("Kernel_Name", ""): continue
        tot[row["Counter_Name"]][0] += float(row["Counter_Value"]); tot[row["Counter_Name"]][1] += 1
w = tot["SQ_WAVES"][0] / max(tot["SQ_WAVES"][1], 1)
for c in sorted(tot):
    s, n = tot[c]
    print(f"{c:22s} per-wave {s/n/w:10.1f}")
PY
