#!/bin/bash
# Round-end evidence on the GPU box: bench line, rocprofv3 kernel stats, HBM traffic counters.
# usage (through gpurun): bash profiles/final_profile.sh <tag>
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p $OUT
cd $REPO
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --stationary --no-cpu-baseline > $OUT/bench_stationary.json 2>> $OUT/bench.err; echo "bench stationary rc=$?"
timeout -k 10 300 python3 bench.py --reverb --steps 256 --warmup 128 > $OUT/bench_reverb.json 2>> $OUT/bench.err; echo "bench reverb rc=$?"
timeout -k 10 120 python3 profiles/latency.py > $OUT/latency.txt 2>> $OUT/bench.err; echo "latency rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 64 --warmup 200 --no-cpu-baseline > $OUT/trace.log 2>&1; echo "trace rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_reverb -- python3 $REPO/bench.py --reverb --steps 64 --warmup 128 --no-cpu-baseline > $OUT/trace_reverb.log 2>&1; echo "trace reverb rc=$?"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_$N.log 2>&1; echo "pmc $N rc=$?"
done
python3 - <<PY
import csv, glob, collections, json, os
out = "$OUT"
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        name = "fused" if ("fused_group_kernel" in k or "fused_block_kernel" in k) else "mix" if "mix_kernel" in k else "prep" if "prep_kernel" in k else None
        if not name: continue
        t = tot[name][row["Counter_Name"]]
        t[0] += float(row["Counter_Value"]); t[1] += 1
summ = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in tot.items()}
f = summ.get("fused", {})
# MI355X_MICROARCH.md (HBM): FETCH_SIZE is in KB and reads 1/2 of a wide coalesced stream on gfx950 -> x2;
# WRITE_SIZE (KB) reads exact for 16-B-per-lane stores.
if "FETCH_SIZE" in f and "WRITE_SIZE" in f:
    hbm = f["FETCH_SIZE"] * 1024 * 2 + f["WRITE_SIZE"] * 1024
    json.dump({"hbm_bytes_per_launch": hbm, "fetch_kb_raw": f["FETCH_SIZE"], "write_kb_raw": f["WRITE_SIZE"],
               "note": "fused_group_kernel, 64 blocks x 1024 moving sources per launch; FETCH_SIZE x2 (gfx950 "
                       "wide-load correction, MI355X_MICROARCH.md), separate --pmc passes"},
              open(out + "/traffic.json", "w"), indent=1)
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1)
for k, d in summ.items():
    for c in sorted(d): print(f"{k:6s} {c:28s} {d[c]:16.1f}")
PY
cat $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -8
cat $OUT/trace_reverb/*/*kernel_stats.csv 2>/dev/null | head -8
cat $OUT/latency.txt
