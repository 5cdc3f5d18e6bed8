#!/bin/bash
# Which stage owns the LDS bank conflicts of fused_pair_kernel?  (13.6 % of LDS-active cycles.)
#  1. PMC of the debug kernels that run the same device code in isolation (profiles/lds_attribution.py)
#  2. PMC of the fused kernel as shipped and with the table-based distance factor switched off
#     (variant TABLE0 = -DJF_TABLE_DISTANCE=0: minimax sin/cos, no LDS gather)
# usage (through gpurun): bash profiles/lds_attribution.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/lds_attr
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/debug -- python3 $REPO/profiles/lds_attribution.py > $OUT/debug.log 2>&1; echo "debug kernels rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/product -- python3 $REPO/bench.py --pmc-child > $OUT/product.log 2>&1; echo "product rc=$?"
JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_TABLE0.so timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/table0 -- python3 $REPO/bench.py --pmc-child > $OUT/table0.log 2>&1; echo "table0 rc=$?"
python3 - <<PY
import csv, glob, collections
for tag in ("debug", "product", "table0"):
    tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob("$OUT/" + tag + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "").split("(")[0][-40:]
            t = tot[k][row["Counter_Name"]]
            t[0] += float(row["Counter_Value"]); t[1] += 1
    for k, d in sorted(tot.items()):
        if not any(n in k for n in ("debug", "fused")): continue
        v = {c: s / n for c, (s, n) in d.items()}
        w = max(v.get("SQ_WAVES", 1), 1)
        conf, act, ins = v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_INSTS_LDS", 0)
        print(f"{tag:8s} {k:42s} launches {d['SQ_WAVES'][1]:4d} per wave: LDS insts {ins / w:9.1f} active {act / w:10.1f} "
              f"conflict {conf / w:9.1f} = {conf / max(act, 1):6.3f} of active; addr-conflict {v.get('SQ_LDS_ADDR_CONFLICT', 0) / w:8.1f}")
PY
