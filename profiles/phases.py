"""Where does a wave of fused_pair_kernel spend its time?  Needs a library built with -DJF_EXP_PHASES
(make -C jefferson-2.0_amd/csrc variant TAG=PHASES KFLAGS=-DJF_EXP_PHASES; JF_LIB=.../libjefferson_hip_PHASES.so python
profiles/phases.py).  Runs the bench's launch shape a few times and prints, per phase, the share of the waves' cycles
(mean over the waves of the first 64 workgroups; shader-clock cycles from s_memtime)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jf_load import jf  # noqa: E402
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("wl", os.path.join(ROOT, "jefferson-2.0_amd", "workload.py"))
wl = importlib.util.module_from_spec(spec)
spec.loader.exec_module(wl)
hrir = np.load(os.path.join(ROOT, "tests", "golden", "kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
S, KB, B = 1024, 128, 256
ids = np.arange(S)
e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=KB)
for s in ids:
    e.set_signal(int(s), wl.source_signal_and_start(s)[0])
n_pos = 5760
e.upload_positions(wl.trajectories(jf, ids, n_pos))
for i in range(40):
    e.batch_run((i * KB) % n_pos, KB)
e.synchronize()
raw = (C.c_ulonglong * 4096)()
assert jf.lib().jf_debug_read_stamps(e.h, raw, 4096) == 0
u = np.frombuffer(raw, np.uint32)[: 64 * 16 * 8].reshape(64, 16, 8).astype(np.float64)
e.close()
names = ["own: descriptor, records, window requests", "own: window arrival + forward transform + distance factors",
         "own: hand-off (slot wait, mailbox stores)", "own: two half-filters", "partner: descriptor/row requests + two half-filters",
         "end of unit: exchange, inverse, crossfade, store", "start of unit: scan, wait for partner's reads",
         "partner: waiting for his hand-off"]
tot = u.sum(axis=2).mean()
print(f"cycles per wave and launch (mean of {u.shape[0] * u.shape[1]} waves): {tot:.0f}")
for k, n in enumerate(names):
    v = u[:, :, k]
    print(f"  {100 * v.mean() / tot:5.1f} %  {v.mean():9.0f} cycles  (waves 0-7: {v[:, :8].mean():9.0f}, waves 8-15: {v[:, 8:].mean():9.0f})  {n}")
