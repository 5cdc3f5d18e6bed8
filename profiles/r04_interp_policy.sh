#!/bin/bash
# Round 4: the pre-interpolated rows against the per-block weighting by how often the sources move (JF_INTERP_TABLE=0 never,
# 1 always), and what the per-run default (2) picks.  One box, one call.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_policy
mkdir -p $OUT
run() { # tag, bench args (quoted), env...
  tag=$1; args=$2; shift 2
  env "$@" python3 bench.py $args --steps 512 --cpu-sample-blocks 132 > $OUT/$tag.json 2> $OUT/$tag.err || { echo "$tag FAILED"; tail -5 $OUT/$tag.err; }
  python3 - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d["roofline"].get("hbm", {})
    print("%-14s value %.4e  step %.4f ms  launch %.4f ms  verified %s  rows %s  hbm %s GB/launch  valu/sb %s" % (
        sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("verified"),
        d["config"]["interp_table"]["rows_read_by_the_timed_runs"],
        ("%.3f" % (h["bytes_per_launch"] / 1e9)) if h.get("bytes_per_launch") else "-",
        ("%.0f" % d["roofline"]["issue"]["valu_insts_per_source_block"]) if "issue" in d["roofline"] else "-"))
except Exception as ex:
    print(sys.argv[2], "no line:", ex)
PY
}
for rep in 1 2; do
  for me in 1 2 4 172; do
    run me${me}_never_$rep  "--no-pmc --move-every $me" JF_INTERP_TABLE=0
    run me${me}_always_$rep "--no-pmc --move-every $me" JF_INTERP_TABLE=1
  done
done
run default_pmc "" JF_X=1
run me2_auto    "--no-pmc --move-every 2" JF_X=1
run me172_pmc   "--move-every 172" JF_X=1
run stat_auto   "--no-pmc --stationary" JF_X=1
