#!/bin/bash
# Round 6: the spatialiser's pair kernel behind the reverb stage -- fused_pair_kernel<2>, G = 16, B = 128,
# 65 536 source-blocks per launch -- beside the headline's <4> (G = 32, B = 256, 131 072 per launch): counters per source-block.
# usage: profiles/r06_pair2_pmc.sh <tag> [lib-tag]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pair2_${1:-x}
[ -n "$2" ] && export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$2.so
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for CFG in head reverb; do
  [ $CFG = reverb ] && A="--reverb" || A=""
  i=0
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_VALU_FLOPS_FP32" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$CFG.p$i -- python3 $REPO/bench.py --pmc-child $A > $OUT/$CFG.p$i.log 2>&1
    echo "$CFG pass $i rc=$?"
    i=$((i+1))
  done
done
python3 - <<PY
import csv, glob, collections
out = "$OUT"
res = {}
for cfg, items in (("head", 1024 * 128), ("reverb", 256 * 256)):
    tot = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/{cfg}.p*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "fused_pair_kernel" not in row.get("Kernel_Name", ""): continue
            t = tot[row["Counter_Name"]]; t[0] += float(row["Counter_Value"]); t[1] += 1
    res[cfg] = {c: s / n / items for c, (s, n) in tot.items()}
with open(out + "/summary.txt", "w") as o:
    hdr = "%-30s %14s %14s %8s" % ("per source-block", "<4> head", "<2> reverb", "ratio")
    print(hdr); o.write(hdr + "\n")
    for c in sorted(set(res["head"]) | set(res["reverb"])):
        a, b = res["head"].get(c, 0.0), res["reverb"].get(c, 0.0)
        line = "%-30s %14.3f %14.3f %8.3f" % (c, a, b, b / a if a else 0.0)
        print(line); o.write(line + "\n")
PY
