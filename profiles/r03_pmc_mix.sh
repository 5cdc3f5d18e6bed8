#!/bin/bash
# instruction mix and fetch counters of the fused kernel: bash profiles/r03_pmc_mix.sh [lib-tag]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_mix_${1:-product}
[ -n "$1" ] && export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$1.so
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
         "SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM" \
         "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_FLOPS_FP32 SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM"; do
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
        line = f"{c:28s} per-launch {s/n:16.0f}  per-item {s/n/ITEMS:10.2f}"
        print(line); o.write(line + "\n")
PY
