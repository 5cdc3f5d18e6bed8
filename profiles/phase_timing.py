"""Per-phase shader-clock durations of one work item inside the fused kernel under full load
(experiment build: make variant TAG=timing KFLAGS=-DJF_EXP_TIMING; run with JF_LIB=...timing.so).
The stamps overwrite the output blocks, so this is a profile run only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util  # noqa: E402

from jf_load import jf  # noqa: E402

spec = importlib.util.spec_from_file_location("jf_workload", os.path.join(ROOT, "jefferson-2.0_amd", "workload.py"))
wl = importlib.util.module_from_spec(spec)
spec.loader.exec_module(wl)
S, K, B = 1024, 64, 256
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
eng = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=K)
for s in range(S):
    eng.set_signal(s, wl.source_signal_and_start(s)[0])
G = int(os.environ.get("JF_SOURCE_GROUP", "4"))
eng.set_source_group(G)
pos = wl.trajectories(jf, np.arange(S), 3 * K, moving=True)
names = ["gather", "rfft", "D*X", "filt old", "mirror old", "ifft old", "filt new", "mirror new", "ifft new", "xfade", "store"]
for rep in range(3):
    eng.process_batch(pos[rep * K:(rep + 1) * K])
raw = eng.read_device(eng.partial_device_ptr(), (K * (S // G), 2 * B))
ts = raw.view(np.uint64)[:, :12].astype(np.int64)
d = np.diff(ts, axis=1)
ok = (d >= 0).all(axis=1) & (d < 10_000_000).all(axis=1)
d = d[ok]
tot = (ts[ok, 11] - ts[ok, 0])
print(f"units {len(d)} of {len(ts)}; last item of each unit (G = {G}); shader-clock ticks")
for n, col in zip(names, d.T):
    print(f"  {n:11s} mean {col.mean():9.0f}  median {np.median(col):9.0f}  p90 {np.percentile(col, 90):9.0f}")
print(f"  {'item total':11s} mean {tot.mean():9.0f}  median {np.median(tot):9.0f}")
