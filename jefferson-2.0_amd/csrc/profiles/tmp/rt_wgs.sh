#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/rt_wgs; mkdir -p $OUT
export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_m1s.so
for R in 1 2; do
  for W in 256 128 64 32 512; do
    JF_RV_SIDE_WGS=$W timeout -k 10 200 python3 $REPO/profiles/latency_reverb.py > $OUT/w${W}_$R.txt 2>&1
    echo "== m1s wgs=$W $R"; head -n 2 $OUT/w${W}_$R.txt | sed 's/configs.*jf_process_block//'
  done
done
