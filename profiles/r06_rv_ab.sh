#!/bin/bash
# Round 6: A/B of the batch reverb's kernels under rocprofv3 (config 5 at the bench's shape): profiles/r06_rv_ab.sh <tag> ...
# ("-" = the product library; a tag = libjefferson_hip_<tag>.so from `make variant`).  Prints the average duration of every
# reverb kernel and of the step; every line of bench.py verifies itself against the oracle.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_rv_ab
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS=${RV_ARGS:---reverb --no-pmc --no-cpu-baseline --steps 64 --warmup 8}
for T in "$@"; do
  if [ "$T" = "-" ]; then unset JF_LIB; N=product; else export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$T.so; N=$T; fi
  rm -rf $OUT/$N
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$N -- python3 $REPO/bench.py $ARGS > $OUT/$N.json 2> $OUT/$N.err
  echo "== $N rc=$?"
  python3 - "$OUT/$N" "$OUT/$N.json" <<'PY'
import csv, glob, json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("   value %.4e  ms_per_step %.4f  verified %s" % (d["value"], d["ms_per_step"], d.get("verified")))
except Exception as ex:
    print("   no bench line:", ex)
tot = 0.0
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "reverb" in r["Name"] or "fused" in r["Name"] or "mix" in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows:
        if int(r["Calls"]) < 10: continue
        n = r["Name"].split("(")[0].replace("void jf::", "")
        print("   %-44s calls %5s avg %8.2f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
        if "reverb" in n: tot += float(r["AverageNs"]) / 1e3
print("   reverb kernels together %.1f us" % tot)
PY
done
