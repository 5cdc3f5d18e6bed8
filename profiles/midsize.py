import os, sys, time
import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from jf_load import jf
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
rng = np.random.default_rng(1)
for S, K in ((64, 64), (256, 16), (32, 128)):
    e = jf.Engine(256, 512, S, hrir=hrir, max_batch_blocks=K)
    for s in range(S):
        e.set_signal(s, rng.uniform(-0.5, 0.5, 44100).astype(np.float32))
    pos = np.zeros((K, S, 5), np.float32)
    for k in range(K):
        for s in range(S):
            pos[k, s] = jf.position_from_spherical(-30 + (7 * s) % 100, (37 * s + k) % 360, 1.0)
    for _ in range(20):
        e.process_batch(pos)
    e.upload_positions(pos)
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        e.batch_run(0, K, None)
    e.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"S={S} K={K}: {dt*1e6:.1f} us per call, {S*K*256/dt:.3e} source-frames/s")
    e.close()
