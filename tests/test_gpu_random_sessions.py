"""Random sessions: long seeded sequences of everything a host can do to an engine -- setters (spherical and cartesian, whole
and fractional degrees, positions outside the measured range), new signals (longer and shorter than a block, empty), resets,
pause, the FD_BASIC switch, per-block calls, batch calls of ragged sizes, the callback's one-block-late ordering -- applied
to the HIP engine and to the C oracle in lockstep, every block compared.  The engine-only knobs (real-time kernel or batch
pipeline for per-block calls, the pre-interpolated rows, the source grouping, the reverb's side stream and partitioning) are
flipped in mid-session as well: they must never change a result beyond float32 rounding.  What the scenario tests check one
at a time is checked here in the orders nobody thought of."""
import numpy as np
import pytest

import oracle_lib
from conftest import sum_tol

pytestmark = pytest.mark.gpu

TOL32 = 4e-7


def _signal(rng, castanets):
    kind = rng.integers(0, 5)
    if kind == 0:
        return np.zeros(0, np.float32)                                   # empty: silence
    if kind == 1:
        n = int(rng.integers(1, 200))                                    # shorter than a block: wraps several times in it
    else:
        n = int(rng.integers(300, 9000))
    a = int(rng.integers(0, len(castanets) - n))
    return (0.4 * castanets[a:a + n] + 0.05 * rng.uniform(-1, 1, n)).astype(np.float32)


def _move(rng, eng, ora, s, log=None):
    kind = rng.integers(0, 4)
    if kind == 0:
        ele, azi, r = float(rng.integers(-40, 91)), float(rng.integers(0, 360)), float(rng.uniform(0.2, 3.0))
    elif kind == 1:
        ele, azi, r = float(rng.uniform(-60, 100)), float(rng.uniform(-20, 380)), float(rng.uniform(0.05, 6.0))  # fractional, outside
    # The engine's setters refuse what the index/weight rule cannot place (elevations outside (-50, 90]: JF_ERR_RANGE, the
    # source stays where it was) -- the reference would read its tables out of bounds there; the oracle is only told what
    # the engine accepted.
    if kind <= 1:
        rc = eng.set_spherical(s, ele, azi, r)
        if rc == 0:
            ora.set_spherical(s, ele, azi, r)
        else:
            assert rc == -2 and not (-50 < round(ele) <= 90), (rc, ele)       # JF_ERR_RANGE
        if log is not None:
            log.append(f"sph s{s} {ele:.3f} {azi:.3f} {r:.3f} rc={rc}")
    else:
        x, y, z = (float(v) for v in rng.uniform(-2, 2, 3))
        if abs(x) + abs(y) + abs(z) < 0.05:
            z = 1.0
        rc = eng.set_cartesian(s, x, y, z)
        if rc == 0:
            ora.set_cartesian(s, x, y, z)
        if log is not None:
            log.append(f"cart s{s} {x:.3f} {y:.3f} {z:.3f} rc={rc}")


@pytest.mark.parametrize("seed,B,S,reverb", [(1, 256, 5, 0), (2, 128, 9, 0), (3, 128, 4, 2500), (4, 128, 3, 16 * 128 * 3 + 77), (5, 64, 6, 0),
                                                 (6, 192, 4, 0), (7, 256, 40, 0), (8, 256, 6, 8 * 256 * 4 + 5), (9, 256, 7, 0)])
def test_random_session_of_block_and_batch_calls(jf, hrir, castanets, seed, B, S, reverb):
    rng = np.random.default_rng(1000 + seed)
    corrected = seed == 9        # one session under JF_FLAG_CORRECTED_INTERPOLATION (the oracle's mode bit 1)
    eng = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=24, flags=jf.JF_FLAG_CORRECTED_INTERPOLATION if corrected else 0)
    ora = oracle_lib.Engine(B, 512, S, hrir)
    ora.set_mode(2 if corrected else 0)
    for s in range(S):
        sig = _signal(rng, castanets) if s else (0.4 * castanets[:7000]).astype(np.float32)
        eng.set_signal(s, sig)
        ora.set_signal(s, sig)
    P = 0
    if reverb:
        ir = (rng.standard_normal(reverb) * np.exp(-4.0 * np.arange(reverb) / reverb)).astype(np.float32)
        ir /= np.sqrt((ir ** 2).sum())
        eng.set_reverb(ir, 0.5)
        ora.set_reverb(ir, 0.5)
        P = -(-reverb // B)
    tol_rel = sum_tol(TOL32 + (2e-7 + 1e-7 * np.sqrt(P) if P else 0.0), S)   # S sources: their errors add like sqrt(S)
    worst = peak = 0.0
    blocks = 0
    paused = False
    log = []
    for step in range(170):
        op = rng.integers(0, 100)
        if op < 30:
            for s in rng.integers(0, S, int(rng.integers(1, S + 1))):
                _move(rng, eng, ora, int(s), log)
        elif op < 36:
            s = int(rng.integers(0, S))
            sig = _signal(rng, castanets)
            eng.set_signal(s, sig)
            ora.set_signal(s, sig)
            log.append(f"signal s{s} n={len(sig)}")
        elif op < 40:
            s = int(rng.integers(0, S))
            eng.reset(s)
            ora.reset(s)
            log.append(f"reset s{s}")
        elif op < 44:
            m = int(rng.integers(0, 2))
            eng.set_mode(m)
            ora.set_mode(m | (2 if corrected else 0))
            log.append(f"mode {m}")
        elif op < 48:
            paused = not paused
            eng.set_pause(paused)
            log.append(f"pause {paused}")
        elif op < 53:      # engine-only knobs: results must not care
            k = rng.integers(0, 5)
            log.append(f"knob {int(k)}")
            if k == 0:
                eng.set_rt_max_sources(int(rng.choice([0, 2, 8192])))
            elif k == 1:
                eng.set_interp_table(int(rng.integers(0, 3)))
            elif k == 2:
                eng.set_source_group(int(rng.choice([1, 1, S])) if S % 2 else int(rng.choice([1, 2, S])))
            elif k == 3 and reverb:
                eng.set_reverb_async(bool(rng.integers(0, 2)))
            elif k == 4:
                eng.set_prep_ahead(bool(rng.integers(0, 2)))
        # then always some audio
        if rng.random() < 0.25 and not paused:
            K = int(rng.integers(1, 25))
            pos = np.zeros((K, S, 5), np.float32)
            cur = [eng.get_position(s)[[0, 1, 3, 4, 5]] for s in range(S)]     # {ele, azi, r, x, y, z} -> the latched record
            for k in range(K):
                for s in range(S):
                    u = rng.random()
                    if u < 0.3:
                        cur[s] = jf.position_from_spherical(float(rng.integers(-40, 91)), float(rng.integers(0, 360)), float(rng.uniform(0.2, 3.0)))
                    elif u < 0.38:
                        # a record need not come from a setter: fractional degrees, now and then a position the rule cannot
                        # place (a silent item; the block after it fades in from nothing)
                        ele = float(rng.uniform(-49.4, 90.4)) if u < 0.36 else float(rng.choice([-55.0, 93.5]))
                        cur[s] = np.array([ele, float(rng.uniform(0, 359.9)), *rng.uniform(-2, 2, 3)], np.float32)
                    pos[k, s] = cur[s]
            got = eng.process_batch(pos)
            want = ora.process_batch(pos)
            log.append(f"batch K={K}")
        else:
            got = eng.process_block()[None]
            want = np.zeros_like(got) if paused else ora.process_block()[None]
            log.append("block " + ";".join(eng.last_kernels()[-2:]))
        blocks += len(got)
        peak = max(peak, float(np.abs(want).max()))
        err = float(np.abs(got - want).max())
        worst = max(worst, err)
        if err > tol_rel * max(1.0, float(np.abs(want).max())):
            print("\n".join(log[-40:]))     # (pytest shows it with the failure)
        assert err <= tol_rel * max(1.0, float(np.abs(want).max())), (seed, step, int(op), err)
    eng.close()
    ora.close()
    assert blocks > 200 and peak > 0.02, (blocks, peak)


@pytest.mark.parametrize("seed,B,S,reverb", [(11, 256, 3, 0), (12, 128, 20, 0), (13, 128, 5, 16 * 128 * 3 + 9)])
def test_random_session_through_the_callback(jf, hrir, castanets, seed, B, S, reverb):
    """jf_callback hands out the block submitted by the PREVIOUS call (Audio.cu:104-117): the oracle's block k against the
    engine's call k + 1, with setters, resets, new signals, the mode switch and pause falling between the calls."""
    rng = np.random.default_rng(2000 + seed)
    eng = jf.Engine(B, 512, S, hrir=hrir)
    ora = oracle_lib.Engine(B, 512, S, hrir)
    for s in range(S):
        sig = (0.4 * castanets[1000 * s:1000 * s + 6000]).astype(np.float32)
        eng.set_signal(s, sig)
        ora.set_signal(s, sig)
    tol = sum_tol(TOL32, S)
    if reverb:      # the big partitions' work goes to the side stream a whole big block ahead (16 blocks: 220 calls cross 13 of them)
        ir = (rng.standard_normal(reverb) * np.exp(-4.0 * np.arange(reverb) / reverb)).astype(np.float32)
        ir /= np.sqrt((ir ** 2).sum())
        eng.set_reverb(ir, 0.5)
        ora.set_reverb(ir, 0.5)
        tol = sum_tol(TOL32 + 2e-7 + 1e-7 * np.sqrt(-(-reverb // B)), S)
    prev = np.zeros(2 * B, np.float32)        # intermediate[] before the first block
    paused = False
    peak = 0.0
    for step in range(220):
        op = rng.integers(0, 100)
        if op < 40:
            for s in rng.integers(0, S, 2):
                _move(rng, eng, ora, int(s))
        elif op < 45:
            s = int(rng.integers(0, S))
            sig = _signal(rng, castanets)
            eng.set_signal(s, sig)
            ora.set_signal(s, sig)
        elif op < 49:
            s = int(rng.integers(0, S))
            eng.reset(s)
            ora.reset(s)
        elif op < 53:
            m = int(rng.integers(0, 2))
            eng.set_mode(m)
            ora.set_mode(m)
        elif op < 57:
            paused = not paused
            eng.set_pause(paused)
        got = eng.callback()
        assert np.abs(got - prev).max() <= tol * max(1.0, float(np.abs(prev).max())), (seed, step, int(op), float(np.abs(got - prev).max()), float(np.abs(prev).max()), float(np.abs(got).max()), paused)
        prev = np.zeros(2 * B, np.float32) if paused else ora.process_block()
        peak = max(peak, float(np.abs(prev).max()))
    rc, last = eng.collect_block()
    assert rc == 0 and np.abs(last - prev).max() <= tol * max(1.0, float(np.abs(prev).max()))
    eng.close()
    ora.close()
    assert peak > 0.02


@pytest.mark.parametrize("seed,B,S,shards,reverb", [(21, 256, 7, 3, 0), (22, 128, 10, 2, 0), (23, 128, 6, 3, 16 * 128 * 4 + 1)])
def test_random_session_through_the_group_of_shards(jf, hrir, castanets, seed, B, S, shards, reverb):
    """The same through include/jefferson_group.h with several shards of the job on the one device (production code for the
    sharding, the repack of the positions, the routing of per-source calls by GLOBAL index, the job-wide controls; the wire
    replaced by a host sum): one job, the C oracle beside it."""
    import importlib
    group = importlib.import_module("jefferson_amd.group")       # needs libjefferson_group.so (RCCL at build time)
    rng = np.random.default_rng(3000 + seed)
    g = group.Group(B, 512, S, hrir, max_batch_blocks=12, shards_on_device=shards)
    assert g.num_gpus() == shards
    ora = oracle_lib.Engine(B, 512, S, hrir)
    for s in range(S):
        sig = (0.4 * castanets[900 * s:900 * s + 5000 + 31 * s]).astype(np.float32)
        g.set_signal(s, sig)
        ora.set_signal(s, sig)
    tol = sum_tol(TOL32, S)
    if reverb:
        ir = (rng.standard_normal(reverb) * np.exp(-4.0 * np.arange(reverb) / reverb)).astype(np.float32)
        ir /= np.sqrt((ir ** 2).sum())
        g.set_reverb(ir, 0.5)
        ora.set_reverb(ir, 0.5)
        tol = sum_tol(TOL32 + 2e-7 + 1e-7 * np.sqrt(-(-reverb // B)), S)
    paused = False
    peak = 0.0
    blocks = 0
    for s in range(S):
        g.set_spherical(s, 0.0, float(20 * s), 1.0)
        ora.set_spherical(s, 0.0, float(20 * s), 1.0)
    for step in range(120):
        op = rng.integers(0, 100)
        if op < 35:
            for s in rng.integers(0, S, 3):
                _move(rng, g, ora, int(s))
        elif op < 40:
            s = int(rng.integers(0, S))
            sig = _signal(rng, castanets)
            g.set_signal(s, sig)
            ora.set_signal(s, sig)
        elif op < 44:
            s = int(rng.integers(0, S))
            g.reset(s)
            ora.reset(s)
        elif op < 48:
            m = int(rng.integers(0, 2))
            assert g.set_mode(m) == 0, "jf_group_set_mode"
            ora.set_mode(m)
        elif op < 52:
            paused = not paused
            g.set_pause(paused)
        if rng.random() < 0.3 and not paused:
            K = int(rng.integers(1, 20))       # beyond max_batch_blocks: processed in runs
            pos = np.zeros((K, S, 5), np.float32)
            for k in range(K):
                for s in range(S):
                    pos[k, s] = jf.position_from_spherical(float(rng.integers(-40, 91)), float(rng.integers(0, 360)), float(rng.uniform(0.2, 3.0)))
            got = g.process_batch(pos)
            want = ora.process_batch(pos)
        else:
            got = g.process_block()[None]
            want = np.zeros_like(got) if paused else ora.process_block()[None]
        blocks += len(got)
        peak = max(peak, float(np.abs(want).max()))
        assert np.abs(got - want).max() <= tol * max(1.0, float(np.abs(want).max())), (seed, step, int(op))
    assert not g.failed() and blocks > 100 and peak > 0.02, (g.failed(), blocks, peak)
    g.close()
    ora.close()


@pytest.mark.parametrize("seed,B,S,G", [(31, 256, 16, 4), (32, 128, 12, 0), (33, 256, 8, 1)])
def test_random_session_of_runs_over_an_uploaded_trajectory(jf, hrir, castanets, seed, B, S, G):
    """jf_batch_upload_positions / jf_batch_run: windows of an uploaded trajectory in and out of order -- a run that follows
    its predecessor finds its descriptors prepared by it (inside the pair kernel's launch or the mix kernel's), any other run
    must not -- with everything that may invalidate them in between: another upload, the mode switch, the rows' policy, a
    reset, per-block calls, a new grouping.  Against the oracle fed the same windows."""
    rng = np.random.default_rng(4000 + seed)
    K = 8
    eng = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=K)
    ora = oracle_lib.Engine(B, 512, S, hrir)
    if G:
        eng.set_source_group(G)
    for s in range(S):
        sig = (0.4 * castanets[700 * s:700 * s + 6000 + 13 * s]).astype(np.float32)
        eng.set_signal(s, sig)
        ora.set_signal(s, sig)

    def trajectory(n):
        ele = rng.integers(-40, 91, S).astype(np.float32)
        azi0 = rng.integers(0, 360, S)
        every = rng.integers(1, 5, S)           # some sources move every block, some dwell
        b = np.arange(n)[:, None]
        azi = (azi0[None, :] + b // every[None, :]) % 360
        return jf.positions_from_spherical(np.broadcast_to(ele, (n, S)), azi.astype(np.float32),
                                           np.broadcast_to(rng.uniform(0.3, 2.5, S).astype(np.float32), (n, S)))
    pos = trajectory(64)
    eng.upload_positions(pos)
    nxt = 0
    peak = 0.0
    prepared = 0
    for step in range(60):
        op = rng.integers(0, 100)
        if op < 8:
            pos = trajectory(int(rng.integers(2, 9)) * K)
            eng.upload_positions(pos)
            nxt = 0
        elif op < 14:
            m = int(rng.integers(0, 2))
            eng.set_mode(m)
            ora.set_mode(m)
        elif op < 20:
            eng.set_interp_table(int(rng.integers(0, 3)))
        elif op < 25:
            s = int(rng.integers(0, S))
            eng.reset(s)
            ora.reset(s)
        elif op < 30 and G != 1:
            eng.set_source_group(int(rng.choice([1, 2, 4])))
        elif op < 35:
            eng.set_prep_ahead(bool(rng.integers(0, 2)))
        elif op < 45:                            # a per-block call in between (positions: where the last run left the sources)
            a, b = eng.process_block(), ora.process_block()
            assert np.abs(a - b).max() <= sum_tol(TOL32, S) * max(1.0, float(np.abs(b).max())), (seed, step, "block")
        # a run: mostly the window that follows, sometimes any other
        n = int(rng.integers(1, K + 1))
        if rng.random() < 0.25 or nxt + n > len(pos):
            first = int(rng.integers(0, len(pos) - n + 1))
        else:
            first = nxt
        eng.batch_run(first, n)
        eng.synchronize()
        prepared += any("prep" in k and k != "prep_kernel" for k in eng.last_kernels())
        got = eng.read_device(eng.mix_device_ptr(), (n, 2 * B))
        want = ora.process_batch(pos[first:first + n])
        eng.set_latched(pos[first + n - 1])      # jf_batch_run leaves the sources alone: say where they stand
        nxt = first + n
        peak = max(peak, float(np.abs(want).max()))
        assert np.abs(got - want).max() <= sum_tol(TOL32, S) * max(1.0, float(np.abs(want).max())), (seed, step, int(op), first, n, eng.last_kernels())
    eng.close()
    ora.close()
    assert peak > 0.02 and prepared >= 1, (peak, prepared)
