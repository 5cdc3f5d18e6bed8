#!/bin/bash
# real-time config 5: product library against variants, alternating processes: rt_ab.sh <tag> ...
# (WGS_<tag>=n in the environment: JF_RV_SIDE_WGS=n for that tag's runs)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/rt_ab; mkdir -p $OUT
for R in 1 2; do
  for T in "$@"; do
    if [ "$T" = "-" ]; then unset JF_LIB; N=product; else export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$T.so; N=$T; fi
    V=WGS_$N; if [ -n "${!V}" ]; then export JF_RV_SIDE_WGS=${!V}; else unset JF_RV_SIDE_WGS; fi
    timeout -k 10 200 python3 $REPO/profiles/latency_reverb.py > $OUT/${N}_$R.txt 2>&1
    echo "== $N $R"; head -n 2 $OUT/${N}_$R.txt
  done
done
for T in "$@"; do
  if [ "$T" = "-" ]; then unset JF_LIB; N=product; else export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$T.so; N=$T; fi
  V=WGS_$N; if [ -n "${!V}" ]; then export JF_RV_SIDE_WGS=${!V}; else unset JF_RV_SIDE_WGS; fi
  timeout -k 10 200 python3 $REPO/bench.py --reverb --realtime --steps 1000 --warmup 300 --no-pmc > $OUT/bench_$N.json 2> $OUT/bench_$N.err
  python3 -c "
import json
d=json.loads(open('$OUT/bench_$N.json').read().strip().splitlines()[-1])
print('bench $N value %.4e verified %s' % (d['value'], d.get('verified')), {k: round(v,1) for k,v in d['realtime_call_us'].items() if k in ('mean','median','p99','max')})"
done
cd /tmp; export TMPDIR=/tmp
for T in "$@"; do
  if [ "$T" = "-" ]; then unset JF_LIB; N=product; else export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$T.so; N=$T; fi
  V=WGS_$N; if [ -n "${!V}" ]; then export JF_RV_SIDE_WGS=${!V}; else unset JF_RV_SIDE_WGS; fi
  rm -rf $OUT/trace_$N
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$N -- python3 $REPO/bench.py --reverb --realtime --steps 500 --warmup 300 --no-pmc --no-cpu-baseline > $OUT/trace_$N.log 2>&1
  echo "== trace $N"; cat $OUT/trace_$N/*/*kernel_stats.csv | grep -E "big_mac|big_fft|big_ifft" | cut -d, -f1-4,6,7 | cut -c1-160
done
