#!/bin/bash
# the run's mix inside the pair kernel's launch against a launch of its own, same library, alternating: bash profiles/r03_mix_ab.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for R in 1 2 3; do
  for M in 0 1; do
    JF_MIX_IN_FUSED=$M timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-pmc --no-cpu-baseline > gpurun_out/mixab_${M}_$R.json 2> gpurun_out/mixab_${M}_$R.err || echo "FAILED $M $R"
    python3 - <<PY
import json
d = json.load(open("gpurun_out/mixab_${M}_$R.json"))
print("mix in fused = $M: value %.4g  step %.4f ms  fused launch %.4f ms  outside %.1f us  verified %s  kernels %s" % (
    d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], 1e3 * (d["ms_per_step"] - d["roofline"]["avg_launch_ms"]), d.get("verified"), d.get("kernels_last_step", d.get("kernels"))))
PY
  done
done
