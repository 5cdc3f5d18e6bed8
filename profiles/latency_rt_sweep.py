"""Per-block calls with many sources: the one-launch real-time kernel against the prep -> fused -> mix pipeline
(jf_debug_set_rt_max_sources picks), median host-to-host latency of jf_process_block (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jf_load import jf
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
sig = (np.load(os.path.join(ROOT, "tests/golden/castanets_441_excerpt_i24.npy")) / 8388608.0).astype(np.float32)
L = jf.lib()
for S in (4096, 8192, 16384):
    res = []
    for rt_max in (0, 32768):
        e = jf.Engine(256, 512, S, hrir=hrir)
        L.jf_debug_set_rt_max_sources(e.h, rt_max)
        for s in range(S):
            e.set_signal(s, sig)
        out = np.zeros(512, np.float32)
        fp = out.ctypes.data_as(jf._f)
        for k in range(50):
            L.jf_process_block(e.h, fp)
        ts = []
        for k in range(300):
            if k % 4 == 0:
                for s in range(0, S, 7):
                    e.set_spherical(s, 5, (3 + k + s) % 360, 0.5)
            t0 = time.perf_counter()
            L.jf_process_block(e.h, fp)
            ts.append(time.perf_counter() - t0)
        res.append(np.median(np.array(ts)) * 1e6)
        e.close()
    print(f"S={S:5d}: pipeline {res[0]:6.1f} us, one-launch kernel {res[1]:6.1f} us", flush=True)
