cd /root/repo
for G in 32 64 128; do
  JF_SOURCE_GROUP=$G python3 bench.py --steps 300 --warmup 20 --no-pmc --no-cpu-baseline > gpurun_out/g$G.json 2> gpurun_out/g$G.err || echo FAILED $G
  python3 profiles/bench_brief.py gpurun_out/g$G.json
done
python3 bench.py --steps 300 --warmup 20 --no-pmc --no-cpu-baseline > gpurun_out/gauto.json 2>/dev/null; python3 profiles/bench_brief.py gpurun_out/gauto.json
