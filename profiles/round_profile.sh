#!/bin/bash
# Round evidence on the GPU box: bench lines (default / stationary / reverb / reverb real-time), rocprofv3 kernel
# stats of the same commands, PMC passes of the fused kernel.   usage (through gpurun): bash profiles/round_profile.sh <tag>
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p $OUT
cd $REPO
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2>> $OUT/bench.err; echo "bench as the driver runs it rc=$?"
timeout -k 10 300 python3 bench.py --stationary > $OUT/bench_stationary.json 2>> $OUT/bench.err; echo "bench stationary rc=$?"
timeout -k 10 300 python3 bench.py --move-every 172 > $OUT/bench_move_every_172.json 2>> $OUT/bench.err; echo "bench move-every 172 rc=$?"
timeout -k 10 300 python3 bench.py --reverb --steps 256 --warmup 128 > $OUT/bench_reverb.json 2>> $OUT/bench.err; echo "bench reverb rc=$?"
timeout -k 10 300 python3 bench.py --reverb --realtime --steps 2000 --warmup 500 > $OUT/bench_reverb_realtime.json 2>> $OUT/bench.err; echo "bench reverb realtime rc=$?"
# the same kernel with a delay line larger than the 256 MiB Infinity Cache: the HBM-bound measurement
timeout -k 10 300 python3 bench.py --reverb --realtime --rv-sources 512 --steps 2000 --warmup 500 > $OUT/bench_reverb_realtime_512src_hbm.json 2>> $OUT/bench.err; echo "bench reverb realtime 512 sources rc=$?"
timeout -k 10 300 python3 bench.py --reverb --realtime --rv-ir-seconds 4.0 --steps 2000 --warmup 500 > $OUT/bench_reverb_realtime_4s_hbm.json 2>> $OUT/bench.err; echo "bench reverb realtime 4 s IR rc=$?"
# a long run (5.1e6 blocks x 1024 sources, verified at its end) and the N > 1 code path rehearsed with two gloo ranks on this one GPU
timeout -k 10 300 python3 bench.py --steps 20000 --warmup 256 --no-pmc > $OUT/bench_soak_20000_steps.json 2>> $OUT/bench.err; echo "bench soak rc=$?"
JF_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --steps 64 --warmup 16 > $OUT/bench_2rank_gloo_rehearsal.json 2>> $OUT/bench.err; echo "bench 2-rank gloo rehearsal rc=$?"
# the C host's number (jefferson_group.h, one GPU) and configs[0]/[1] through the plain-C offline driver
timeout -k 10 120 ./jefferson-2.0_amd/jf_ctest bench 1 512 > $OUT/ctest_bench.txt 2>> $OUT/bench.err; echo "jf_ctest bench rc=$?"
timeout -k 10 300 python3 profiles/render_config1.py > $OUT/render_config1.txt 2>> $OUT/bench.err; echo "render config1 rc=$?"
timeout -k 10 120 python3 profiles/latency.py > $OUT/latency.txt 2>> $OUT/bench.err; echo "latency rc=$?"
timeout -k 10 120 python3 profiles/latency_reverb.py > $OUT/latency_reverb.txt 2>> $OUT/bench.err; echo "latency reverb rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 64 --warmup 200 --no-cpu-baseline --no-pmc --no-also > $OUT/trace.log 2>&1; echo "trace rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_reverb -- python3 $REPO/bench.py --reverb --steps 64 --warmup 128 --no-pmc > $OUT/trace_reverb.log 2>&1; echo "trace reverb rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_reverb_rt -- python3 $REPO/bench.py --reverb --realtime --steps 500 --warmup 300 --no-pmc > $OUT/trace_reverb_rt.log 2>&1; echo "trace reverb rt rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_reverb_rt512 -- python3 $REPO/bench.py --reverb --realtime --rv-sources 512 --steps 500 --warmup 300 --no-pmc > $OUT/trace_reverb_rt512.log 2>&1; echo "trace reverb rt 512 rc=$?"
cd $REPO
bash profiles/quick_pmc.sh $TAG > $OUT/pmc.txt 2>&1; echo "pmc rc=$?"
cp gpurun_out/qpmc_$TAG/summary.txt $OUT/pmc_summary.txt 2>/dev/null
for d in trace trace_reverb trace_reverb_rt trace_reverb_rt512; do echo "== $d"; cat $OUT/$d/*/*kernel_stats.csv 2>/dev/null | head -8; done
python3 profiles/bench_brief.py $OUT/bench.json $OUT/bench_stationary.json
cat $OUT/latency.txt $OUT/latency_reverb.txt $OUT/ctest_bench.txt $OUT/render_config1.txt
