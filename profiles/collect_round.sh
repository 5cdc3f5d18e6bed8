#!/bin/bash
# copies the summaries of gpurun_out/final_<tag>/ (written by round_profile.sh on the GPU box) into profiles/<tag>/
TAG=${1:-r04}
cd "$(dirname "$0")/.."
S=gpurun_out/final_$TAG; D=profiles/$TAG
mkdir -p $D
for f in bench bench_driver_shape bench_stationary bench_move_every_172 bench_reverb bench_reverb_realtime bench_reverb_realtime_512src_hbm bench_reverb_realtime_4s_hbm bench_soak_20000_steps bench_2rank_gloo_rehearsal; do
  [ -s $S/$f.json ] && tail -n 1 $S/$f.json > $D/$f.json
done
sed -i "/^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl/d" $S/ctest_bench.txt 2>/dev/null
cp $S/latency.txt $S/latency_reverb.txt $S/ctest_bench.txt $S/render_config1.txt $S/pmc_summary.txt $D/ 2>/dev/null
cp "$(ls -t $S/trace/*/*kernel_stats.csv | head -n 1)" $D/kernel_stats.csv 2>/dev/null   # the newest run's
cp "$(ls -t $S/trace_reverb/*/*kernel_stats.csv | head -n 1)" $D/kernel_stats_reverb.csv 2>/dev/null   # the newest run's
cp "$(ls -t $S/trace_reverb_rt/*/*kernel_stats.csv | head -n 1)" $D/kernel_stats_reverb_realtime.csv 2>/dev/null   # the newest run's
cp "$(ls -t $S/trace_reverb_rt512/*/*kernel_stats.csv | head -n 1)" $D/kernel_stats_reverb_realtime_512src_hbm.csv 2>/dev/null   # the newest run's
ls -la $D
