"""Which blocks of a run of calls does the non-uniformly partitioned reverb get wrong?  (How the fut-ring collision of round 4 was found:
runs of calls of several shapes against the float64 model, per-block errors.)  usage: python profiles/reverb_calls_debug.py [B]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from jf_load import jf
import model64
from scipy.signal import fftconvolve
gold = os.path.join(ROOT, "tests", "golden")
hrir = np.load(os.path.join(gold, "kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
cast = (np.load(os.path.join(gold, "castanets_441_excerpt_i24.npy")) / 8388608.0).astype(np.float32)
def ir_(n, seed=99, decay=3.0):
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(n) * np.exp(-decay * np.arange(n) / n)
    return (h / np.sqrt((h ** 2).sum())).astype(np.float32)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = 16 if B <= 128 else 8
B1 = M * B
S, K = 1, 70
ir = ir_(3 * B1 + 1000)
sig = cast[:12000]
pos = np.zeros((K, S, 5), np.float32)
for b in range(K):
    pos[b, 0] = jf.position_from_spherical(0, 30, 0.5)
mod = model64.Model(B, 512, S, hrir)
reps = -(-K * B // len(sig))
stream = np.tile(sig.astype(np.float64), reps)[:K * B]
mod.src[0].buf = 0.6 * fftconvolve(stream, ir.astype(np.float64))[:K * B]
mod.src[0].count = 0
want, _ = mod.process_batch(pos)
for name, calls in (("perblock", [1] * K), ("1,5,16,17,31", [1, 5, 16, 17, 31]), ("70", [70]), ("6,64", [6, 64]), ("8,8,8,46", [8, 8, 8, 46]), ("16x4+6", [16, 16, 16, 16, 6])):
    e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=max(calls))
    e.set_reverb_partitioning(2)
    e.set_signal(0, sig)
    e.set_reverb(ir, 0.6)
    got, b0 = [], 0
    kern = []
    for k in calls:
        got.append(e.process_batch(pos[b0:b0 + k]))
        kern.append(";".join(x for x in e.last_kernels() if x.startswith("reverb")))
        b0 += k
    got = np.concatenate(got)
    err = np.abs(got - want).max(axis=1)
    bad = np.nonzero(err > 1e-5)[0]
    print(name, "max err %.3e" % err.max(), "bad blocks", bad.tolist()[:40])
    if len(bad) and len(calls) < 8:
        for c, kk in zip(calls, kern):
            print("   call", c, kk)
    e.close()
