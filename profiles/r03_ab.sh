#!/bin/bash
# A/B of variant builds only (no tests): bash profiles/r03_ab.sh <tag> ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
AB_ARGS="${AB_ARGS:---steps 200 --warmup 20 --no-pmc --no-cpu-baseline}" bash profiles/ab.sh "$@"
