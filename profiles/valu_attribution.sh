#!/bin/bash
# Vector instructions per stage: PMC of the debug kernels that run the fused kernel's device code in isolation
# (profiles/lds_attribution.py: forward transform / distance factors / distance factors + transform + weighted spectra).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/valu_attr
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/debug -- python3 $REPO/profiles/lds_attribution.py > $OUT/debug.log 2>&1; echo "debug kernels rc=$?"
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("$OUT/debug/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "").split("(")[0][-30:]
        if "debug" not in k: continue
        rows[(k, int(row["Dispatch_Id"]))][row["Counter_Name"]] = float(row["Counter_Value"])
for (k, d), v in sorted(rows.items(), key=lambda t: t[0][1]):
    w = max(v.get("SQ_WAVES", 1), 1)
    print(f"{k:32s} dispatch {d:3d} per wave: " + "  ".join(f"{c[3:]} {v[c] / w:8.1f}" for c in sorted(v) if c != "SQ_WAVES"))
PY
