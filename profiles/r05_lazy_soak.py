"""Soak of the put-off small transforms (jf_engine.cpp: rv_small_stale): two engines on one GPU, the same calls -- one that
puts the small transforms of calls of whole big blocks off until somebody needs them, one that forms them at once -- must give
the same blocks BIT FOR BIT: batch calls whose sizes are mostly multiples of the big block (so that many end on a boundary),
ragged ones that move the phase, one-block calls, resets, new signals, moves."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jf_load import jf
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
rng = np.random.default_rng(int(os.environ.get("JF_SOAK_SEED", "7")))
S, B, N = int(os.environ.get("JF_SOAK_SOURCES", "24")), int(os.environ.get("JF_SOAK_BLOCK", "128")), int(os.environ.get("JF_SOAK_CALLS", "3000"))
M = 16 if B <= 128 else 8
n_ir = M * B * 6 + 55
ir = rng.standard_normal(n_ir) * np.exp(-5.0 * np.arange(n_ir) / n_ir)
ir = (ir / np.sqrt((ir ** 2).sum())).astype(np.float32)
sigs = [rng.uniform(-.5, .5, 5000 + 37 * s).astype(np.float32) for s in range(S)]
engines = []
for lazy in (True, False):
    e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=4 * M)
    e.set_reverb_lazy_state(lazy)
    for s in range(S):
        e.set_signal(s, sigs[s])
        e.set_spherical(s, -40 + (s * 7) % 121, (s * 37) % 360, 1.0)
    e.set_reverb(ir, 0.5)
    engines.append(e)
blocks = calls = caught = 0
peak = 0.0
t0 = time.time()
while calls < N:
    r = rng.random()
    if r < 0.45:
        # a batch call: mostly whole big blocks, or what is left up to the next boundary, sometimes anything
        q = rng.random()
        to_boundary = (M - blocks % M) % M
        k = int(rng.integers(1, 4)) * M if q < 0.5 else (to_boundary + int(rng.integers(0, 3)) * M if q < 0.8 and to_boundary else int(rng.integers(2, 4 * M + 1)))
        k = max(1, min(k, 4 * M))
        pos = jf.positions_from_spherical(np.broadcast_to(rng.integers(-40, 90, S).astype(np.float32), (k, S)),
                                          rng.integers(0, 360, (k, S)).astype(np.float32), np.ones((k, S), np.float32))
        y = [e.process_batch(pos) for e in engines]
        blocks += k
    else:
        if r < 0.55:
            for s in rng.integers(0, S, 3):
                a, el = float(rng.integers(0, 360)), float(rng.integers(-40, 90))
                for e in engines:
                    e.set_spherical(int(s), el, a, 1.0)
        y = [e.process_block() for e in engines]
        blocks += 1
    caught += any(k.endswith("@ring") for k in engines[0].last_kernels())
    assert not any(k.endswith("@ring") for k in engines[1].last_kernels())
    calls += 1
    peak = max(peak, float(np.abs(y[1]).max()))
    if not np.array_equal(y[0], y[1]):
        print(f"MISMATCH at call {calls} (block {blocks}): max diff {np.abs(y[0] - y[1]).max():.3e}", engines[0].last_kernels(), flush=True)
        sys.exit(1)
    q = rng.random()
    if q < 0.01:
        s = int(rng.integers(0, S))
        for e in engines:
            e.reset(s)
    elif q < 0.02:
        s = int(rng.integers(0, S))
        sig = rng.uniform(-.5, .5, int(rng.integers(100, 9000))).astype(np.float32)
        for e in engines:
            e.set_signal(s, sig)
print(f"B = {B}, {S} sources: {calls} calls = {blocks} blocks identical bit for bit, {caught} catch-ups, peak {peak:.3f}, {time.time() - t0:.1f} s")
