#!/bin/bash
# Real-time kernel: sources (= waves) per workgroup 16 / 8 / 4 (csrc: -DJF_RT_WAVES).  Per-block latency without and with the
# reverb stage; run on the GPU box from the repo root after `make variant TAG=rt8 KFLAGS=-DJF_RT_WAVES=8` (and rt4).
OUT=gpurun_out/rt_waves
mkdir -p $OUT
for v in "" rt8 rt4; do
  if [ -z "$v" ]; then unset JF_LIB; tag=rt16; else export JF_LIB=$PWD/jefferson-2.0_amd/libjefferson_hip_$v.so; tag=$v; fi
  echo "== $tag" >> $OUT/latency.txt
  python3 profiles/latency.py >> $OUT/latency.txt 2>&1 || exit 1
  JF_RV_ONLY_NONUNIFORM=1 python3 profiles/latency_reverb.py >> $OUT/latency.txt 2>&1 || exit 1
done
cat $OUT/latency.txt
