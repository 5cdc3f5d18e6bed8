#!/bin/bash
# Round 4: how many of 16 moving sources should read pre-interpolated rows (HBM) instead of weighting the cached measured
# rows (vector instructions)?  JF_INTERP_SHARE = 0 .. 16 on the bench workload, the stationary variant with and without the
# rows, and the counters (FETCH_SIZE x 2 + WRITE_SIZE etc., collected by bench.py itself) at share 0 and 16.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_share
mkdir -p $OUT
run() { # tag, bench args (quoted), env...
  tag=$1; args=$2; shift 2
  env "$@" python3 bench.py $args --steps 512 --cpu-sample-blocks 132 > $OUT/$tag.json 2> $OUT/$tag.err || { echo "$tag FAILED"; tail -5 $OUT/$tag.err; }
  python3 - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d["roofline"].get("hbm", {})
    print("%-12s value %.4e  step %.4f ms  launch %.4f ms  verified %s  hbm %s GB/launch  valu/sb %s" % (
        sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("verified"),
        ("%.3f" % (h["bytes_per_launch"] / 1e9)) if h.get("bytes_per_launch") else "-",
        ("%.0f" % d["roofline"]["issue"]["valu_insts_per_source_block"]) if "issue" in d["roofline"] else "-"))
except Exception as ex:
    print(sys.argv[2], "no line:", ex)
PY
}
for rep in 1 2; do
  for sh in 0 4 8 10 12 14 16; do
    run share${sh}_$rep "--no-pmc" JF_INTERP_SHARE=$sh
  done
  run stat_off_$rep "--no-pmc --stationary" JF_INTERP_TABLE=0
  run stat_on_$rep  "--no-pmc --stationary" JF_X=1
done
run pmc_share0  "" JF_INTERP_SHARE=0
run pmc_share16 "" JF_INTERP_SHARE=16
run pmc_share8  "" JF_INTERP_SHARE=8
