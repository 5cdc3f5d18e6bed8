"""Per-block latency of the real-time path (BASELINE.json configs[1]: 1 source, 256-sample blocks)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jf_load import jf
if not os.environ.get("JF_NO_PIN"):
    jf.pin_thread_to_device(0)   # the audio thread on the GPU's NUMA node (include/jefferson.h; profiles/r04/rt_numa.md)
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
sig = (np.load(os.path.join(ROOT, "tests/golden/castanets_441_excerpt_i24.npy")) / 8388608.0).astype(np.float32)
for S in [int(x) for x in os.environ.get("JF_LAT_SOURCES", "1,8,64,256").split(",")]:
    e = jf.Engine(256, 512, S, hrir=hrir)
    for s in range(S):
        e.set_signal(s, sig)
    out = np.zeros(512, np.float32)
    L = jf.lib()
    fp = out.ctypes.data_as(jf._f)
    for k in range(50):
        L.jf_process_block(e.h, fp)
    ts = []
    for k in range(500):
        if k % 4 == 0:
            for s in range(S):
                e.set_spherical(s, 5, (3 + k + s) % 360, 0.5)   # crossfade every 4th block
        t0 = time.perf_counter()
        L.jf_process_block(e.h, fp)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print(f"S={S}: jf_process_block median {np.median(ts):.1f} us  p99 {np.percentile(ts,99):.1f} us  min {ts.min():.1f} us")
    e.close()
