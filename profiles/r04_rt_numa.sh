#!/bin/bash
# Per-block latency by where the calling thread runs: pinned to the GPU's NUMA node by the library's helper (default of the
# scripts), left alone (JF_NO_PIN=1: the system's choice, one of two states per process), and forced onto the other node.
export JF_LAT_SOURCES=1,64,256
python3 -c "
import sys; sys.path.insert(0, '.')
from jf_load import jf
print('GPU 0 is on NUMA node', jf.device_numa_node(0))"
OTHER=$(python3 -c "
import sys; sys.path.insert(0, '.')
from jf_load import jf
n = jf.device_numa_node(0)
print(open('/sys/devices/system/node/node%d/cpulist' % (1 - n)).read().strip() if n in (0, 1) else '')")
for i in 1 2 3; do
  echo "== pinned to the GPU's node (jf_pin_thread_to_device)"; python3 profiles/latency.py || exit 1; JF_RV_ONLY_NONUNIFORM=1 python3 profiles/latency_reverb.py || exit 1
  echo "== left alone (JF_NO_PIN=1)"; JF_NO_PIN=1 python3 profiles/latency.py || exit 1; JF_NO_PIN=1 JF_RV_ONLY_NONUNIFORM=1 python3 profiles/latency_reverb.py || exit 1
  if [ -n "$OTHER" ]; then
    echo "== on the other node (taskset -c $OTHER, JF_NO_PIN=1)"; JF_NO_PIN=1 taskset -c $OTHER python3 profiles/latency.py || exit 1; JF_NO_PIN=1 JF_RV_ONLY_NONUNIFORM=1 taskset -c $OTHER python3 profiles/latency_reverb.py || exit 1
  fi
done
