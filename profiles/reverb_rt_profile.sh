#!/bin/bash
# The one HBM-bound kernel of the path: the real-time form of the reverb's multiply-accumulate stage
# (reverb_mac_kernel<128,1>: 256 sources x 690 KB of delay line per block).  rocprofv3 kernel stats + HBM counters of
#   bench.py --reverb --realtime   (one 128-sample block per call)
# usage (through gpurun): bash profiles/reverb_rt_profile.sh <tag>
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/reverb_rt_$TAG
mkdir -p $OUT
cd $REPO
timeout -k 10 300 python3 bench.py --reverb --realtime --steps 2000 --warmup 500 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --reverb --realtime --steps 500 --warmup 300 --no-pmc > $OUT/trace.log 2>&1; echo "trace rc=$?"
for C in FETCH_SIZE "WRITE_SIZE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-20)
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $REPO/bench.py --reverb --realtime --steps 40 --warmup 10 --no-pmc > $OUT/pmc_$N.log 2>&1; echo "pmc $N rc=$?"
done
python3 - <<PY
import csv, glob, collections, json
out = "$OUT"
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        name = "mac" if "reverb_mac" in k else "fft" if "reverb_fft" in k else None
        if not name: continue
        t = tot[name][row["Counter_Name"]]
        t[0] += float(row["Counter_Value"]); t[1] += 1
summ = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in tot.items()}
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1)
for k, d in summ.items():
    for c in sorted(d): print(f"{k:4s} {c:24s} {d[c]:16.1f}")
PY
cat $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -8
python3 $REPO/profiles/bench_brief.py $OUT/bench.json 2>/dev/null | head -2
python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print(json.dumps(d['reverb_roofline'], indent=1)); print(d['ms_per_step'])"
