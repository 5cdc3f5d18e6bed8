#!/bin/bash
# Round 4, VERDICT item 1: the pre-interpolated integer-degree HRTF rows, A/B on ONE box in ONE call.
#   base      product kernel without the rows (JF_INTERP_TABLE=0): the round-3 path
#   pre       rows from the 386 MB table (HBM): "(b) the latency price"
#   narrow0   round-3 path, every source at elevation 5 (control for the narrow workload)
#   narrow1   rows, every source at elevation 5: 360 rows = 2.9 MB stay in the caches: "(a) the compute bound"
#   touch     rows + the next rows touched a source ahead (variant build -DJF_PRE_TOUCH=1)
# Every line is verified against the C oracle.  Output: gpurun_out/r04_interp/*.json
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_interp
mkdir -p $OUT
run() { # tag, env...
  tag=$1; shift
  env "$@" python3 bench.py --no-pmc --steps 512 --cpu-sample-blocks 132 > $OUT/$tag.json 2> $OUT/$tag.err || { echo "$tag FAILED"; tail -5 $OUT/$tag.err; }
  python3 - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-9s value %.4e  step %.4f ms  launch %.4f ms  verified %s  interp %s" % (
        sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("verified"), d["config"].get("interp_table")))
except Exception as ex:
    print(sys.argv[2], "no line:", ex)
PY
}
TOUCH=$PWD/jefferson-2.0_amd/libjefferson_hip_touch.so
for rep in 1 2; do
  run base_$rep    JF_INTERP_TABLE=0
  run pre_$rep     JF_X=1
  run touch_$rep   JF_LIB=$TOUCH
  run narrow0_$rep JF_INTERP_TABLE=0 JF_BENCH_NARROW=1
  run narrow1_$rep JF_BENCH_NARROW=1
  run ntouch_$rep  JF_BENCH_NARROW=1 JF_LIB=$TOUCH
done
