"""The checker checked: seeded random sessions (setters of every kind incl. positions outside the range, signals shorter than a
block and empty ones, resets, the FD_BASIC switch and the corrected rule, per-block and batch calls) through the float32 C
oracle and the float64 NumPy model in lockstep.  No GPU.  Tolerance: the reference's own end-to-end figure, 2e-7 (absolute for
|y| <= 1: precision_test.cu:2158), per source."""
import numpy as np
import pytest

import model64
import oracle_lib


def _session(seed, B, S, corrected, hrir, castanets):
    rng = np.random.default_rng(seed)
    ora = oracle_lib.Engine(B, 512, S, hrir)
    mod = model64.Model(B, 512, S, hrir)
    base = 2 if corrected else 0
    ora.set_mode(base)
    mod.mode = base
    for s in range(S):
        n = int(rng.choice([0, 37, 900, 5000])) if s else 5000
        a = int(rng.integers(0, 20000))
        sig = (0.4 * castanets[a:a + n]).astype(np.float32)
        ora.set_signal(s, sig)
        mod.set_signal(s, sig)
    worst = peak = 0.0
    for step in range(120):
        op = rng.integers(0, 100)
        if op < 45:
            for s in rng.integers(0, S, 2):
                s = int(s)
                if rng.random() < 0.6:
                    ele = float(rng.integers(-40, 91)) if rng.random() < 0.8 else float(rng.uniform(-60, 100))
                    azi, r = float(rng.uniform(-10, 370)), float(rng.uniform(0.1, 4.0))
                    ora.set_spherical(s, ele, azi, r)
                    mod.set_spherical(s, ele, azi, r)
                else:
                    x, y, z = (float(v) for v in rng.uniform(-2, 2, 3))
                    if abs(x) + abs(y) + abs(z) < 0.05:
                        z = 1.0
                    ora.set_cartesian(s, x, y, z)
                    mod.set_cartesian(s, x, y, z)
        elif op < 52:
            s = int(rng.integers(0, S))
            ora.reset(s)
            mod.reset(s)
        elif op < 58:
            m = int(rng.integers(0, 2)) | base
            ora.set_mode(m)
            mod.mode = m
        elif op < 64:
            s = int(rng.integers(0, S))
            n = int(rng.choice([0, 11, 300, 3000]))
            a = int(rng.integers(0, 20000))
            sig = (0.4 * castanets[a:a + n]).astype(np.float32)
            ora.set_signal(s, sig)
            mod.set_signal(s, sig)
        if rng.random() < 0.3:
            K = int(rng.integers(1, 6))
            pos = np.zeros((K, S, 5), np.float32)
            for k in range(K):
                for s in range(S):
                    ele = float(rng.uniform(-49.4, 90.9)) if rng.random() < 0.9 else float(rng.choice([-55.0, 95.0]))
                    pos[k, s] = [ele, float(rng.uniform(0, 359.9)), *rng.uniform(-2, 2, 3)]
            got = ora.process_batch(pos)
            want, _ = mod.process_batch(pos)
        else:
            got, want = ora.process_block()[None], mod.process_block()[None]
        peak = max(peak, float(np.abs(want).max()))
        worst = max(worst, float(np.abs(got - want).max()))
        assert np.abs(got - want).max() <= 2e-7 * S * max(1.0, float(np.abs(want).max())), (seed, step, int(op))
    ora.close()
    return worst, peak


@pytest.mark.parametrize("seed,B,S,corrected", [(1, 256, 3, False), (2, 128, 4, False), (3, 256, 2, True), (4, 64, 3, False),
                                                  (5, 192, 3, False), (6, 128, 5, True), (7, 256, 1, False), (8, 128, 2, False)])
def test_oracle_and_float64_model_in_lockstep(hrir, castanets, seed, B, S, corrected):
    worst, peak = _session(seed, B, S, corrected, hrir, castanets)
    assert peak > 0.01, peak
