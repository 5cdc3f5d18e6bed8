#!/bin/bash
# round 3, first GPU call: the whole GPU suite, then the bench lines of this round's first items
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputests_1.log 2>&1; rc=$?
tail -5 gpurun_out/r03_gputests_1.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 200 --warmup 20 > gpurun_out/r03_bench_1.json 2> gpurun_out/r03_bench_1.err || exit 1
python bench.py --reverb --steps 40 --warmup 5 > gpurun_out/r03_bench_reverb_1.json 2> gpurun_out/r03_bench_reverb_1.err || exit 1
python bench.py --reverb --realtime --steps 2000 --warmup 50 > gpurun_out/r03_bench_reverb_rt_1.json 2> gpurun_out/r03_bench_reverb_rt_1.err || exit 1
python bench.py --reverb --realtime --rv-sources 512 --steps 2000 --warmup 50 > gpurun_out/r03_bench_reverb_rt512_1.json 2> gpurun_out/r03_bench_reverb_rt512_1.err || exit 1
python bench.py --reverb --realtime --rv-ir-seconds 4.0 --steps 2000 --warmup 50 > gpurun_out/r03_bench_reverb_rt4s_1.json 2> gpurun_out/r03_bench_reverb_rt4s_1.err || exit 1
./jefferson-2.0_amd/jf_ctest bench 1 256 > gpurun_out/r03_ctest_bench.txt 2>&1
cat gpurun_out/r03_ctest_bench.txt
