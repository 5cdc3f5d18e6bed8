#!/bin/bash
# Round 4, item 3: config 5 with non-uniform partitioning: batch and real-time lines, kernel stats under rocprofv3.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_reverb
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --reverb > $OUT/bench_reverb.json 2> $OUT/bench_reverb.err; echo "batch rc=$?"
python3 bench.py --reverb --realtime --steps 2048 > $OUT/bench_reverb_rt.json 2> $OUT/bench_reverb_rt.err; echo "rt rc=$?"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_batch -- python3 $OLDPWD/bench.py --reverb --no-pmc --no-cpu-baseline --steps 64 --warmup 8 > $OLDPWD/$OUT/prof_batch.log 2>&1); echo "prof batch rc=$?"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_rt -- python3 $OLDPWD/bench.py --reverb --realtime --no-pmc --no-cpu-baseline --steps 1024 --warmup 64 > $OLDPWD/$OUT/prof_rt.log 2>&1); echo "prof rt rc=$?"
python3 - <<'PY'
import json, glob, csv
for f in ("gpurun_out/r04_reverb/bench_reverb.json", "gpurun_out/r04_reverb/bench_reverb_rt.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value %.4e ms_per_step %.4f verified %s kernels %s" % (d["value"], d["ms_per_step"], d.get("verified"), d["config"]["kernels"]))
        print("   verification", d.get("verification"))
    except Exception as ex:
        print(f, "no line", ex)
for d in ("prof_batch", "prof_rt"):
    for f in glob.glob(f"gpurun_out/r04_reverb/{d}/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
        print(d)
        for r in rows[:9]:
            print("   %-70s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
JF_RV_PARTITIONING=1 python3 bench.py --reverb --no-pmc > $OUT/bench_reverb_uniform.json 2> $OUT/bench_reverb_uniform.err; echo "uniform rc=$?"
python3 - <<'PY'
import json
for f in ("bench_reverb", "bench_reverb_rt", "bench_reverb_uniform"):
    try:
        d = json.loads(open(f"gpurun_out/r04_reverb/{f}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "value %.4e" % d["value"], "stage %.4f ms" % r["avg_stage_ms"], r.get("bound"), "frac %.3f" % r.get("frac", 0), "traffic", r.get("traffic"), "alg bytes", r.get("algorithmic_bytes_per_step"))
        print("   split", d.get("step_split_ms"))
    except Exception as ex:
        print(f, "no line", ex)
PY
