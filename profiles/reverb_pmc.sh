#!/bin/bash
# PMC passes over the batch reverb's multiply-accumulate kernel: profiles/reverb_pmc.sh <tag> [lib-tag]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export JF_REVERB_BLOCKS_PER_STEP=${JF_REVERB_BLOCKS_PER_STEP:-32}  # the size the round's A/B numbers were taken at
OUT=$REPO/gpurun_out/rvpmc_${1:-x}
[ -n "$2" ] && export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$2.so
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
         "FETCH_SIZE" "WRITE_SIZE" \
         "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py --reverb --steps 24 --warmup 8 --no-pmc --no-cpu-baseline > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "reverb_mac_tiled" not in row.get("Kernel_Name", ""): continue
        tot[row["Counter_Name"]][0] += float(row["Counter_Value"]); tot[row["Counter_Name"]][1] += 1
w = tot["SQ_WAVES"][0] / max(tot["SQ_WAVES"][1], 1)
with open("$OUT/summary.txt", "w") as o:
    for c in sorted(tot):
        s, n = tot[c]
        line = f"{c:28s} per-launch {s/n:16.0f}  per-wave {s/n/max(w,1):12.1f}"
        print(line); o.write(line + "\n")
PY
