#!/bin/bash
# A/B of kernel builds on the GPU box: profiles/ab.sh <tag> [<tag> ...]   ("-" = the product library)
# Each tag is a library built with `make -C jefferson-2.0_amd/csrc variant TAG=<tag> KFLAGS=...`.
# Every run keeps its stderr and exit code; a failed variant prints FAILED, never a traceback.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
ARGS=${AB_ARGS:---steps 256 --warmup 256 --no-cpu-baseline --no-pmc}
for T in "$@"; do
  if [ "$T" = "-" ]; then unset JF_LIB; N=product; else export JF_LIB=$REPO/jefferson-2.0_amd/libjefferson_hip_$T.so; N=$T; fi
  for R in 1 2; do
    timeout -k 10 300 python3 bench.py $ARGS > gpurun_out/ab_${N}_$R.json 2> gpurun_out/ab_${N}_$R.err
    RC=$?
    if [ $RC -ne 0 ] || [ ! -s gpurun_out/ab_${N}_$R.json ]; then
      echo "FAILED $N rep $R rc=$RC: $(tail -n 2 gpurun_out/ab_${N}_$R.err | tr '\n' ' ')"
    else
      python3 profiles/bench_brief.py gpurun_out/ab_${N}_$R.json
    fi
  done
done
