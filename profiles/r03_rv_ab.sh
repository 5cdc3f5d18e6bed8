#!/bin/bash
# real-time reverb A/B: bash profiles/r03_rv_ab.sh <tag> ...   ("-" = product)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for T in "$@"; do
  if [ "$T" = "-" ]; then unset JF_LIB; N=product; else export JF_LIB=$PWD/jefferson-2.0_amd/libjefferson_hip_$T.so; N=$T; fi
  for SRC in 256 512; do
    timeout -k 10 200 python3 bench.py --reverb --realtime --rv-sources $SRC --steps 3000 --warmup 100 --no-pmc --no-cpu-baseline > gpurun_out/rvab_${N}_$SRC.json 2> gpurun_out/rvab_${N}_$SRC.err || { echo "FAILED $N $SRC"; continue; }
    python3 - gpurun_out/rvab_${N}_$SRC.json $N $SRC <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(f"{sys.argv[2]:8s} S={sys.argv[3]:4s} step {d['ms_per_step']*1e3:7.2f} us  reverb stage {r['avg_launch_ms']*1e3:7.2f} us  {r['achieved']:7.0f} GB/s  frac {r['frac']:.3f}  {r['level'][:14]}")
PY
  done
done
