#!/bin/bash
# PMC passes over the fused kernel of one build: profiles/quick_pmc.sh <tag> [lib-tag]
# (lib-tag = a variant built with `make variant TAG=...`; empty = the product library)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/qpmc_${1:-x}
[ -n "$2" ] && export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$2.so
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
         "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_WR"; do
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
import os
ITEMS = 1024 * int(os.environ.get("JF_BLOCKS_PER_STEP", "128"))   # source-blocks per launch of bench.py
w = tot["SQ_WAVES"][0] / max(tot["SQ_WAVES"][1], 1)
with open("$OUT/summary.txt", "w") as o:
    for c in sorted(tot):
        s, n = tot[c]
        line = f"{c:28s} per-launch {s/n:14.0f}  per-wave {s/n/w:10.1f}  per-item {s/n/ITEMS:9.2f}"
        print(line); o.write(line + "\n")
PY
