#!/bin/bash
# Kernel stats of the default bench under rocprofv3: profiles/trace_default.sh <tag>   (through gpurun)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_${1:-x}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $REPO/bench.py --steps 64 --warmup 200 --no-cpu-baseline --no-pmc > $OUT/trace.log 2>&1; echo "trace rc=$?"
cat $OUT/t/*/*kernel_stats.csv | head -6 | cut -c1-200
