"""How the CPU baseline (float32 C oracle, OpenMP over sources) scales with threads on the GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle_lib
from jf_load import jf
import importlib.util
spec = importlib.util.spec_from_file_location("wl", os.path.join(ROOT, "jefferson-2.0_amd", "workload.py"))
wl = importlib.util.module_from_spec(spec); spec.loader.exec_module(wl)
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
S, K = 1024, 128
print("affinity cpus:", len(os.sched_getaffinity(0)), "omp max threads:", oracle_lib.lib().jfo_num_threads(), "os.cpu_count:", os.cpu_count())
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cgroup cpu.max: n/a", e)
ora = oracle_lib.Engine(256, 512, S, hrir)
for s in range(S):
    ora.set_signal(s, wl.source_signal_and_start(s)[0])
pos = wl.trajectories(jf, np.arange(S), K)
for nt in (1, 8, 16, 32, 64, 128):
    for s in range(S):
        ora.reset(s)
    t0 = time.perf_counter(); ora.process_batch(pos, n_threads=nt); dt = time.perf_counter() - t0
    print(f"threads {nt:4d}: {S*K*256/dt:.3e} source-frames/s  ({dt:.2f} s)")
