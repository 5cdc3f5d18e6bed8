"""Worst-case output error of the HIP path against the float64 model on loud and quiet material (GPU box).
Prints max |HIP - model64| / max(1, |y|) per case; the reference's bound is 2e-7 (precision_test.cu:2158)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
from jf_load import jf  # noqa: E402
import model64  # noqa: E402

gold = os.path.join(ROOT, "tests", "golden")
hrir = np.load(os.path.join(gold, "kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
cast = (np.load(os.path.join(gold, "castanets_441_excerpt_i24.npy")) / 8388608.0).astype(np.float32)
rng = np.random.default_rng(11)
noise = rng.uniform(-.5, .5, 8192).astype(np.float32)
for name, sig, blocks in (("noise 0.5", noise, 12), ("castanets", cast, 40)):
    for r in (0.05, 0.5, 1.0, 2.0, 3.5, 4.9):
        e = jf.Engine(256, 512, 1, hrir=hrir)
        m = model64.Model(256, 512, 1, hrir)
        for x in (e, m):
            x.set_signal(0, sig)
        worst = peak = sq = 0.0
        for b in range(blocks):
            for x in (e, m):
                x.set_spherical(0, 10 * (b % 3), (45 + 7 * b) % 360, r)
            y, y64 = e.process_block(), m.process_block()
            worst = max(worst, np.abs(y - y64).max() / max(1.0, np.abs(y64).max()))
            peak = max(peak, np.abs(y64).max())
            sq += float(np.mean((y - y64) ** 2))
        e.close()
        print(f"{name:10s} r {r:4.2f}  peak {peak:6.3f}  max err / max(1,|y|) {worst:.3e}  rms err {np.sqrt(sq / blocks):.3e}")
# the stage tap: D against float64, in ulps of its modulus
coords = [(0.0, 0.0, 0.05), (0.5, 0.0, 0.0), (0.3, 0.4, 1.2), (0.0, 3.0, 4.0), (2.0, -1.0, 7.0), (10.0, 5.0, -20.0),
          (60.0, 0.0, 80.0), (0.0, 100.0, 0.0), (57.7, 57.7, 57.8), (1e-3, 0.0, 0.0)]
pos = np.array([[0.0, 0.0, x, y, z] for x, y, z in coords], np.float32)
e = jf.Engine(256, 512, 1, hrir=hrir)
D = e.stage_taps(pos)
e.close()
for i, c in enumerate(coords):
    d64 = model64.distance_factor(c, 513)
    ulp = np.abs(d64[0]) * 2.0 ** -23
    err = np.abs(D[i, :512] - d64[:512]) / ulp
    print(f"D tap {c}: max err {err.max():.2f} ulp of |D|, rms {np.sqrt(np.mean(err ** 2)):.3f}")
