#!/bin/bash
# PMC passes over the batch reverb's big-partition kernels (config 5 at the bench's shape): profiles/r04_reverb_big_pmc.sh <tag> [lib-tag]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/rvbig_${1:-x}
[ -n "$2" ] && export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$2.so
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
         "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py --reverb --steps 12 --warmup 4 --no-pmc --no-cpu-baseline > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "reverb" not in k: continue
        k = k.split("(")[0].replace("void jf::", "")
        t = tot[k][row["Counter_Name"]]
        t[0] += float(row["Counter_Value"]); t[1] += 1
with open("$OUT/summary.txt", "w") as o:
    for k in sorted(tot):
        w = tot[k]["SQ_WAVES"][0] / max(tot[k]["SQ_WAVES"][1], 1)
        o.write("== %s\n" % k); print("==", k)
        for c in sorted(tot[k]):
            s, n = tot[k][c]
            line = f"{c:28s} per-launch {s/n:16.0f}  per-wave {s/n/max(w,1):12.1f}"
            print(line); o.write(line + "\n")
PY
