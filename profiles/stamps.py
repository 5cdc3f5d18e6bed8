"""When do the waves of the batch kernel start and finish?  Needs a library built with -DJF_EXP_STAMPS
(make variant TAG=STAMPS KFLAGS=-DJF_EXP_STAMPS; JF_LIB=.../libjefferson_hip_STAMPS.so python profiles/stamps.py)."""
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
S, KB = 1024, int(os.environ.get("JF_STAMP_BLOCKS", "64"))  # 64: 4096 units = two per resident pair; bench.py runs 128: four
eng = jf.Engine(256, 512, S, hrir=hrir, max_batch_blocks=KB)
ids = np.arange(S)
for s in ids:
    eng.set_signal(int(s), wl.source_signal_and_start(s)[0])
pos = wl.trajectories(jf, ids, 2880)
eng.upload_positions(pos)
for i in range(300):
    eng.batch_run((i % (2880 // KB)) * KB, KB)
eng.synchronize()
st = eng.read_stamps(8192).reshape(2048, 4).astype(np.float64) * 10.0 / 1e3  # us (100 MHz counter); per pair:
t0 = st[:, 0].min()                                                            # start, end of round 0, of round 1, exit
st -= t0
r0, r1 = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1]
print(f"pairs start: max {st[:, 0].max():.1f} us; kernel ends at {st[:, 3].max():.1f} us")
print(f"round 0 unit: median {np.median(r0):.1f}, p10 {np.percentile(r0, 10):.1f}, p90 {np.percentile(r0, 90):.1f}, max {r0.max():.1f} us")
print(f"round 1 unit: median {np.median(r1):.1f}, p10 {np.percentile(r1, 10):.1f}, p90 {np.percentile(r1, 90):.1f}, max {r1.max():.1f} us")
print(f"pair busy (rounds 0-1): mean {np.mean(st[:, 2] - st[:, 0]):.1f} us = {np.mean(st[:, 2] - st[:, 0]) / st[:, 3].max():.2f} of the kernel")
ex = st[:, 3]
print(f"pairs exit: p1 {np.percentile(ex, 1):.1f}, p10 {np.percentile(ex, 10):.1f}, median {np.median(ex):.1f}, p90 {np.percentile(ex, 90):.1f}, "
      f"p99 {np.percentile(ex, 99):.1f}, max {ex.max():.1f} us; mean life {np.mean(ex - st[:, 0]):.1f} us = {np.mean(ex - st[:, 0]) / ex.max():.3f} of the kernel")
blk0 = np.arange(2048) % 64                 # block of the round-0 unit (round 1, zigzag: 63 - blk0)
xcd = (np.arange(2048) // 8) % 8
print("by XCD (workgroup % 8): round-0 median, round-1 median, end median")
for x in range(8):
    m = xcd == x
    print(f"  XCD {x}: {np.median(r0[m]):6.1f} {np.median(r1[m]):6.1f} {np.median(st[m, 2]):6.1f}")
print("by block of the round-0 unit: round-0 median / round-1 median (block 63 - b)")
for b in (0, 1, 2, 3, 4, 8, 16, 32, 48, 56, 59, 60, 61, 62, 63):
    m = blk0 == b
    print(f"  b {b:2d}: {np.median(r0[m]):6.1f} / {np.median(r1[m]):6.1f}")
eng.close()
# who leaves last?
ex = st[:, 3]
pidx = np.arange(2048) % 8
print("exit by pair index inside the workgroup (waves 2i, 2i+1): median / p99 / max")
for i in range(8):
    m = pidx == i
    print(f"  pair {i}: {np.median(ex[m]):6.1f} / {np.percentile(ex[m], 99):6.1f} / {ex[m].max():6.1f}")
late = np.argsort(ex)[-24:]
print("the last 24 pairs (pair id, workgroup, pair in workgroup, first-round block, exit us):")
for p in late:
    print(f"  {p:5d} wg {p // 8:4d} pair {p % 8} block {p % KB:4d} exit {ex[p]:6.1f}  rounds {r0[p]:5.1f} {r1[p]:5.1f}")
bl = np.arange(2048) % KB
print("exit by first-round block (median / max): ", ", ".join(f"b{b}: {np.median(ex[bl == b]):.0f}/{ex[bl == b].max():.0f}" for b in (0, 1, 2, 3, 8, 64, KB - 4, KB - 3, KB - 2, KB - 1)))
