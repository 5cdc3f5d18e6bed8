"""Which source-group size G should jf_batch_run take for a call of S sources x K blocks?  Times every G against the
engine's automatic choice (GPU box): python profiles/group_sweep.py"""
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jf_load import jf  # noqa: E402

hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
rng = np.random.default_rng(1)
for S, K in ((1024, 128), (1024, 64), (1024, 32), (1024, 16), (1024, 8), (1024, 4), (1024, 2), (1024, 1), (256, 64),
             (256, 16), (256, 4), (64, 64), (64, 16), (32, 128), (16, 16)):
    e = jf.Engine(256, 512, S, hrir=hrir, max_batch_blocks=K)
    for s in range(S):
        e.set_signal(s, rng.uniform(-0.5, 0.5, 8192).astype(np.float32))
    sidx = np.arange(S)
    ele = np.broadcast_to((-30 + (7 * sidx) % 100).astype(np.float32), (2 * K, S))
    azi = ((37 * sidx)[None, :] + np.arange(2 * K)[:, None]) % 360
    pos = jf.positions_from_spherical(ele, azi.astype(np.float32), np.full((2 * K, S), 1.0, np.float32))
    e.upload_positions(pos)
    res = {}
    for G in (0, 1, 2, 4, 8, 16, 32):
        if G and S % G:
            continue
        e.set_source_group(G)
        e.upload_positions(pos)   # (the automatic choice also orders the sources)
        for i in range(30):
            e.batch_run((i & 1) * K, K)
        e.synchronize()
        n = 200 if S * K < 65536 else 60
        t0 = time.perf_counter()
        for i in range(n):
            e.batch_run((i & 1) * K, K)
        e.synchronize()
        res[G] = ((time.perf_counter() - t0) / n * 1e6, e.last_source_group())
    best = min((v[0], g) for g, v in res.items() if g)
    print(f"S={S:5d} K={K:4d} items={S * K:7d}: auto -> G={res[0][1]:2d} {res[0][0]:8.1f} us | " +
          " ".join(f"G{g}:{v[0]:.1f}" for g, v in res.items() if g) + f" | best G{best[1]} {best[0]:.1f}", flush=True)
    e.close()
