#!/bin/bash
# the whole GPU suite, then the default bench line and the per-block latencies
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputests.log 2>&1; rc=$?
tail -4 gpurun_out/r03_gputests.log
[ $rc -eq 0 ] || exit $rc
bash profiles/ab.sh - 
python profiles/latency.py
