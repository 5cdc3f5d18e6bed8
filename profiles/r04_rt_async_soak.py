"""Soak of the side-stream form of the real-time reverb (jf_engine.cpp: run_reverb_stage): two engines on one GPU, the same
calls -- one with the big partitions on the second stream, one with everything in line -- must give the same blocks BIT FOR
BIT over tens of thousands of calls: one-block calls back to back and paced (random sleeps, so that the side stream finishes
early, late or in the middle of the following calls), batch calls of random sizes in between, moves, now and then a reset or
a new signal."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jf_load import jf
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
rng = np.random.default_rng(4)
S, B, N = int(os.environ.get("JF_SOAK_SOURCES", "64")), int(os.environ.get("JF_SOAK_BLOCK", "128")), int(os.environ.get("JF_SOAK_CALLS", "20000"))
ir = rng.standard_normal(16 * 128 * 9 + 55) * np.exp(-5.0 * np.arange(16 * 128 * 9 + 55) / (16 * 128 * 9))
ir = (ir / np.sqrt((ir ** 2).sum())).astype(np.float32)
sigs = [rng.uniform(-.5, .5, 5000 + 37 * s).astype(np.float32) for s in range(S)]
engines = []
for on in (True, False):
    e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=40)
    e.set_reverb_partitioning(2)
    e.set_reverb_async(on)
    for s in range(S):
        e.set_signal(s, sigs[s])
        e.set_spherical(s, -40 + (s * 7) % 121, (s * 37) % 360, 1.0)
    e.set_reverb(ir, 0.5)
    engines.append(e)
blocks = calls = side = 0
peak = 0.0
t0 = time.time()
while calls < N:
    r = rng.random()
    if r < 0.02:
        k = int(rng.integers(2, 41))
        pos = jf.positions_from_spherical(np.broadcast_to(rng.integers(-40, 90, S).astype(np.float32), (k, S)),
                                          rng.integers(0, 360, (k, S)).astype(np.float32), np.ones((k, S), np.float32))
        y = [e.process_batch(pos) for e in engines]
        blocks += k
    else:
        if r < 0.1:
            for s in rng.integers(0, S, 5):
                a, el = int(rng.integers(0, 360)), int(rng.integers(-40, 90))
                for e in engines:
                    e.set_spherical(int(s), el, a, 1.0)
        elif r < 0.103:
            s = int(rng.integers(0, S))
            for e in engines:
                e.reset(s)
        elif r < 0.106:
            s = int(rng.integers(0, S))
            sig = rng.uniform(-.5, .5, int(rng.integers(1500, 9000))).astype(np.float32)
            for e in engines:
                e.set_signal(s, sig)
        if rng.random() < 0.05:
            time.sleep(float(rng.uniform(0, 300e-6)))
        y = [e.process_block() for e in engines]
        side += any(k.endswith("@side") for k in engines[0].last_kernels())
        blocks += 1
    calls += 1
    peak = max(peak, float(np.abs(y[0]).max()))
    if not np.array_equal(y[0], y[1]):
        print(f"MISMATCH at call {calls} (block {blocks}): max diff {np.abs(y[0] - y[1]).max():.3e}")
        sys.exit(1)
    if calls % 5000 == 0:
        print(f"{calls} calls, {blocks} blocks, {side} hand-overs to the side stream, peak {peak:.3f}, {time.time() - t0:.0f} s", flush=True)
for e in engines:
    e.close()
print(f"identical bit for bit: B = {B}, {calls} calls, {blocks} blocks, {S} sources, {side} hand-overs to the side stream, peak |y| {peak:.3f}")
