import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
from jf_load import jf
import model64
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
for seed in (11, 12, 13, 14, 15, 16):
    rng = np.random.default_rng(seed)
    sig = rng.uniform(-.5, .5, 8192).astype(np.float32)
    out = []
    for r in (0.05, 0.5):
        e = jf.Engine(256, 512, 1, hrir=hrir); m = model64.Model(256, 512, 1, hrir)
        for x in (e, m):
            x.set_signal(0, sig); x.set_spherical(0, 0, 45, r)
        w = 0
        for _ in range(8):
            y, y64 = e.process_block(), m.process_block()
            w = max(w, np.abs(y - y64).max() / max(1.0, np.abs(y64).max()))
        e.close()
        out.append(w)
    print(seed, " ".join(f"{v:.3e}" for v in out))
