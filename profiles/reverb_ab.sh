#!/bin/bash
# A/B of library variants on the batch reverb (config 5): profiles/reverb_ab.sh <tag> ...   ("-" = the product library)
# Prints the average duration of every reverb kernel under rocprofv3 for each variant.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export JF_REVERB_BLOCKS_PER_STEP=${JF_REVERB_BLOCKS_PER_STEP:-32}  # the size the round's A/B numbers were taken at
for T in "$@"; do
  if [ "$T" = "-" ]; then unset JF_LIB; N=product; else export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$T.so; N=$T; fi
  OUT=$REPO/gpurun_out/rvab_$N
  mkdir -p $OUT
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --reverb --steps 64 --warmup 128 --no-pmc --no-cpu-baseline ${RV_ARGS} > $OUT/trace.log 2>&1
  RC=$?
  if [ $RC -ne 0 ]; then echo "FAILED $N rc=$RC: $(tail -n 2 $OUT/trace.log | tr '\n' ' ')"; continue; fi
  echo "== $N"
  python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "reverb" in r["Name"]:
            print("   %-64s calls %5s avg %8.1f us" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1000))
PY
done
