#!/bin/bash
# plain A/B of bench.py --reverb between the product library and variants: ab_plain.sh <tag> ... (alternating, 3 rounds)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ab_plain; mkdir -p $OUT
for R in 1 2 3; do
  for T in "$@"; do
    if [ "$T" = "-" ]; then unset JF_LIB; N=product; else export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$T.so; N=$T; fi
    timeout -k 10 200 python3 $REPO/bench.py ${RV_ARGS:---reverb --no-pmc --no-cpu-baseline --steps 512 --warmup 64} > $OUT/${N}_$R.json 2> $OUT/${N}_$R.err
    python3 -c "
import json,sys
d=json.loads(open('$OUT/${N}_$R.json').read().strip().splitlines()[-1])
print('$N $R value %.4e ms %.4f stage %.4f verified %s' % (d['value'], d['ms_per_step'], d.get('roofline',{}).get('avg_stage_ms',0), d.get('verified')))"
  done
done
