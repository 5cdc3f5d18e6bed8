#!/bin/bash
# instruction-cache counters of the fused kernel: bash profiles/r03_pmc_icache.sh [lib-tag]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_icache_${1:-product}
[ -n "$1" ] && export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$1.so
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
         "SQ_WAVES SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py --pmc-child > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if not any(n in row.get("Kernel_Name", "") for n in ("fused_pair_kernel", "fused_block_kernel")): continue
        tot[row["Counter_Name"]][0] += float(row["Counter_Value"]); tot[row["Counter_Name"]][1] += 1
ITEMS = 1024 * 128
with open("$OUT/summary.txt", "w") as o:
    for c in sorted(tot):
        s, n = tot[c]
        line = f"{c:32s} per-launch {s/n:16.0f}  per-item {s/n/ITEMS:10.2f}"
        print(line); o.write(line + "\n")
PY
