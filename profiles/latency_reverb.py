"""Real-time (one block per call) cost of configs[4]: 256 sources, B = 128, 2 s IR, reverb + spatialiser."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jf_load import jf
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
rng = np.random.default_rng(99)
ir = rng.standard_normal(88200) * np.exp(-6.9 * np.arange(88200) / 88200.0)
ir = (ir / np.sqrt((ir ** 2).sum())).astype(np.float32)
S, B = 256, 128
e = jf.Engine(B, 512, S, hrir=hrir)
for s in range(S):
    e.set_signal(s, np.random.default_rng(1234 + s).uniform(-.5, .5, 44100).astype(np.float32))
    e.set_spherical(s, -40 + (s * 7) % 121, (s * 37) % 360, 1.0)
e.set_reverb(ir, 0.5)
out = np.zeros(2 * B, np.float32)
fp = out.ctypes.data_as(jf._f)
L = jf.lib()
for k in range(20):
    L.jf_process_block(e.h, fp)
ts = []
for k in range(200):
    t0 = time.perf_counter(); L.jf_process_block(e.h, fp); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print(f"configs[4] real-time: 256 sources, B=128, 690 partitions: jf_process_block median {np.median(ts):.1f} us, "
      f"p99 {np.percentile(ts, 99):.1f} us (block period 2902 us); FDL read per block {S*690*1024/1e6:.0f} MB "
      f"-> {S*690*1024/np.median(ts)/1e6:.2f} TB/s if it were all the time")
e.close()
