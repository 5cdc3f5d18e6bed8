#!/bin/bash
# latency only: rt_ab2.sh <tag[:wgs]> ...   (two alternating rounds)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/rt_ab2; mkdir -p $OUT
for R in 1 2; do
  for X in "$@"; do
    T=${X%%:*}; W=${X#*:}; [ "$W" = "$X" ] && W=""
    if [ "$T" = "-" ]; then unset JF_LIB; N=product; else export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$T.so; N=$T; fi
    if [ -n "$W" ]; then export JF_RV_SIDE_WGS=$W; else unset JF_RV_SIDE_WGS; fi
    timeout -k 10 200 python3 $REPO/profiles/latency_reverb.py > $OUT/${N}_${W}_$R.txt 2>&1
    echo "== $N wgs=${W:-default} $R"; head -n 2 $OUT/${N}_${W}_$R.txt | sed 's/configs.*jf_process_block//'
  done
done
