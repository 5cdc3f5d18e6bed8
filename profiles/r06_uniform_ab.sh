#!/bin/bash
# Round 6: the uniform-partition product kernel (reverb_mac_kernel) with the delay line's spectra as non-temporal loads, A/B under
# rocprofv3 in the three shapes it serves one-block calls in: the 32-partition head of config 5 in real time, the pinned uniform
# form of the 2 s response (690 partitions, 181 MB of delay line: inside the Infinity Cache) and the same with 512 sources
# (362 MB: past it).   usage: profiles/r06_uniform_ab.sh <tag> ...   ("-" = the product library)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_uniform_ab
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for T in "$@"; do
  if [ "$T" = "-" ]; then unset JF_LIB; N=product; else export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$T.so; N=$T; fi
  for SHAPE in head uni690 uni690x512; do
    case $SHAPE in
      head) unset JF_RV_PARTITIONING; A="--reverb --realtime";;
      uni690) export JF_RV_PARTITIONING=1; A="--reverb --realtime";;
      uni690x512) export JF_RV_PARTITIONING=1; A="--reverb --realtime --rv-sources 512";;
    esac
    rm -rf $OUT/$N.$SHAPE
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$N.$SHAPE -- python3 $REPO/bench.py $A --no-pmc --no-cpu-baseline --steps 400 --warmup 100 > $OUT/$N.$SHAPE.json 2> $OUT/$N.$SHAPE.err
    echo "== $N $SHAPE rc=$?"
    python3 - "$OUT/$N.$SHAPE" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ("reverb" in r["Name"] or "rt_block" in r["Name"]) and int(r["Calls"]) >= 100:
            print("   %-60s calls %5s avg %8.2f us" % (r["Name"].split("(")[0].replace("void jf::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
