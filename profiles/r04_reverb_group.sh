#!/bin/bash
# config 5's batch shape with other source-group sizes of the spatialiser (JF_SOURCE_GROUP: bench.py, tuning runs only)
for g in 16 32 8 16 32; do
  JF_SOURCE_GROUP=$g python3 bench.py --reverb --steps 128 --warmup 64 --no-pmc --no-cpu-baseline 2>/dev/null > gpurun_out/rvg_$g.json || exit 1
  python3 - $g <<'PY'
import json, sys
d = json.loads(open("gpurun_out/rvg_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("G", sys.argv[1], "%.4g" % d["value"], "%.4f ms" % d["ms_per_step"], {k: round(v, 4) for k, v in d["step_split_ms"].items()}, d.get("verified"))
PY
done
