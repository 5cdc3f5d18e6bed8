"""Round 4 A/B of the real-time reverb block (configs[4]: 256 sources, B = 128, 2 s response) in ONE process per library and
alternating processes on the same box: the library of the commit before the two-big-block head (JF_LIB=..._prev.so, built
from a worktree of that commit) against the product.  The old library lacks one debug entry: its binding is dropped here
(the product's loader refuses a library with a missing symbol)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jf_load import jf
if os.environ.get("JF_LIB", "").endswith("_prev.so"):
    for name in ("jf_debug_set_reverb_async", "jf_device_numa_node", "jf_pin_thread_to_device"):
        jf._SIGS.pop(name, None)
elif not os.environ.get("JF_NO_PIN"):
    jf.pin_thread_to_device(0)
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
if os.environ.get("JF_RV_ASYNC") == "0":
    e.set_reverb_async(False)
out = np.zeros(2 * B, np.float32)
fp = out.ctypes.data_as(jf._f)
L = jf.lib()
for k in range(64):
    L.jf_process_block(e.h, fp)
ts = []
for k in range(64, 64 + 3200):
    if k % 7 == 0:
        for s in range(0, S, 5):
            e.set_spherical(s, -40 + (s * 7) % 121, (s * 37 + k) % 360, 1.0)
    t0 = time.perf_counter(); L.jf_process_block(e.h, fp); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
n, head, big, taps = e.reverb_partitions()
print(f"{os.path.basename(os.environ.get('JF_LIB', 'product'))} async={os.environ.get('JF_RV_ASYNC', '1')}: {head} x 128 + {big} x {taps}: mean {ts.mean():.1f} us, "
      f"median {np.median(ts):.1f}, p99 {np.percentile(ts, 99):.1f}, max {ts.max():.1f};  cycle medians "
      + " ".join(f"{v:.0f}" for v in np.median(ts.reshape(-1, 16), axis=0)))
e.close()
