import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jf_load import jf
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
sig = (np.load(os.path.join(ROOT, "tests/golden/castanets_441_excerpt_i24.npy")) / 8388608.0).astype(np.float32)
for S in (16, 17, 24, 32, 48, 64, 128):
    for rt in (0, 128):
        e = jf.Engine(256, 512, S, hrir=hrir)
        e.set_rt_max_sources(rt)
        for s in range(S):
            e.set_signal(s, sig)
        out = np.zeros(512, np.float32)
        L = jf.lib()
        fp = out.ctypes.data_as(jf._f)
        for k in range(50):
            L.jf_process_block(e.h, fp)
        ts = []
        for k in range(400):
            if k % 4 == 0:
                for s in range(S):
                    e.set_spherical(s, 5, (3 + k + s) % 360, 0.5)
            t0 = time.perf_counter()
            L.jf_process_block(e.h, fp)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        print(f"S={S} rt_max={rt}: median {np.median(ts):.1f} us  p99 {np.percentile(ts,99):.1f} us")
        e.close()
