#!/bin/bash
# The two states of a real-time process (~3 us apart in every per-block latency): does the CPU the audio thread runs on decide?
# Same script pinned to single CPUs of the box's share, one process each.
export JF_LAT_SOURCES=1,256
nproc; taskset -p $$ ; lscpu | grep -i "numa\|socket\|model name" ; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo
CPUS=$(taskset -p $$ | sed 's/.*: //')
for c in $(python3 -c "
import os
cpus = sorted(os.sched_getaffinity(0))
print(' '.join(str(c) for c in cpus[::max(1, len(cpus)//8)]))"); do
  echo "== cpu $c"; taskset -c $c python3 profiles/latency.py || exit 1
done
echo "== unpinned x3"; for i in 1 2 3; do python3 profiles/latency.py || exit 1; done
