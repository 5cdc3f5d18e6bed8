"""Where the per-block latency of one source goes: the launch (jf_submit_block) and the wait (jf_collect_block) timed
apart on the host (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jf_load import jf
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
sig = (np.load(os.path.join(ROOT, "tests/golden/castanets_441_excerpt_i24.npy")) / 8388608.0).astype(np.float32)
for S in (1, 16, 256):
    e = jf.Engine(256, 512, S, hrir=hrir)
    for s in range(S):
        e.set_signal(s, sig)
    out = np.zeros(512, np.float32)
    L = jf.lib()
    fp = out.ctypes.data_as(jf._f)
    for k in range(100):
        L.jf_process_block(e.h, fp)
    a, b, c = [], [], []
    for k in range(1000):
        t0 = time.perf_counter()
        L.jf_submit_block(e.h)
        t1 = time.perf_counter()
        L.jf_collect_block(e.h, fp)
        t2 = time.perf_counter()
        a.append(t1 - t0); b.append(t2 - t1); c.append(t2 - t0)
    med = lambda v: np.median(np.array(v)) * 1e6
    print(f"S={S}: submit {med(a):.1f} us, collect {med(b):.1f} us, both {med(c):.1f} us")
    e.close()
