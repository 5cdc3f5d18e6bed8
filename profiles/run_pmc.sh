#!/bin/bash
# Collects PMC counters for the fused kernel on the GPU box (separate passes, kernel-trace only).
# usage: profiles/run_pmc.sh <tag>     (run through gpurun; writes gpurun_out/pmc_<tag>/)
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
 "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
)
i=0
for P in "${PASSES[@]}"; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "fused_block_kernel" not in k: continue
        c = row["Counter_Name"]; v = float(row["Counter_Value"])
        tot[c][0] += v; tot[c][1] += 1
with open("$OUT/summary.txt", "w") as o:
    for c in sorted(tot):
        s, n = tot[c]
        line = f"{c:32s} per-launch {s/n:18.1f}   (launches {n})"
        print(line); o.write(line + "\n")
PY
