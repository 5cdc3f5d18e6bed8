#!/bin/bash
# quick GPU check of a kernel change: a test subset, then A/B of variants.  bash profiles/r03_t.sh "<pytest -k expr>" <tag> ...
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
K="$1"; shift
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "$K" > gpurun_out/r03_t.log 2>&1; rc=$?
tail -4 gpurun_out/r03_t.log
[ $rc -eq 0 ] || exit $rc
AB_ARGS="${AB_ARGS:---steps 300 --warmup 20 --no-pmc --no-cpu-baseline}" bash profiles/ab.sh "$@"
