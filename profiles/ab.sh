#!/bin/bash
# A/B of fused-kernel builds on one GPU box: profiles/ab.sh tag1 tag2 ...  ("base" = the product .so)
# extra bench flags via ABFLAGS
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for rep in 1 2; do
for t in "$@"; do
  if [ "$t" = base ]; then L=""; else L="$REPO/jefferson-2.0_amd/libjefferson_hip_$t.so"; fi
  JF_LIB=$L timeout -k 10 120 python3 bench.py --steps 300 --warmup 200 --no-cpu-baseline $ABFLAGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$t rep$rep: value %.3e  step %.4f ms  fused %.4f ms  frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac']))"
done; done
