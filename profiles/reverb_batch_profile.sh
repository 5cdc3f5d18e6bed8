#!/bin/bash
# Batch reverb (config 5) on the GPU box: the bench line and the kernel stats of the same command.
# usage (through gpurun): bash profiles/reverb_batch_profile.sh <tag>
TAG=${1:-rv}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export JF_REVERB_BLOCKS_PER_STEP=${JF_REVERB_BLOCKS_PER_STEP:-32}  # the size the round's A/B numbers were taken at
OUT=$REPO/gpurun_out/reverb_$TAG
mkdir -p $OUT
cd $REPO
timeout -k 10 300 python3 bench.py --reverb --steps 256 --warmup 128 --no-pmc > $OUT/bench_reverb.json 2> $OUT/bench.err; echo "bench reverb rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --reverb --steps 64 --warmup 128 --no-pmc --no-cpu-baseline > $OUT/trace.log 2>&1; echo "trace rc=$?"
cd $REPO
cat $OUT/trace/*/*kernel_stats.csv | head -8
python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_reverb.json').read().strip().splitlines()[-1])
print('value %.4g  ms/step %.4f  verified %s' % (d['value'], d['ms_per_step'], d.get('verified')))
"
