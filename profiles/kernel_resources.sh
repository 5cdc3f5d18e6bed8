#!/bin/bash
# VGPRs / scratch / LDS of every kernel in jf_kernels.hip and jf_reverb.hip as the Makefile builds them (no GPU needed).
cd "$(dirname "$0")/../jefferson-2.0_amd/csrc"
for f in jf_kernels.hip jf_reverb.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $KFLAGS -S --cuda-device-only -o /dev/null $f \
      -Rpass-analysis=kernel-resource-usage 2>&1 |
  sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name:/ {n=$NF} /remark: +VGPRs:/ {v=$NF} /ScratchSize/ {s=$NF} /LDS Size/ {print n, "vgpr", v, "scratch", s, "lds", $NF}' | c++filt
done
