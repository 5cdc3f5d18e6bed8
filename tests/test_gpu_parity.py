"""Parity of the HIP path (through the C ABI) with the oracle, on a real MI355X.

Tolerances.  The reference accepts max |GPU - CPU| <= 2e-7 on its own two float32
paths (precision_test.cu:2158, Precision_Check.py:12) for outputs of magnitude
< 1.  The same bound is used here against the float64 model (TOL64), and twice
that between two float32 paths (HIP vs the C oracle, TOL32), because both carry
their own rounding.
"""
import numpy as np
import pytest

import model64
import oracle_lib
from conftest import scenario_positions

pytestmark = pytest.mark.gpu

TOL64 = 2e-7
TOL32 = 4e-7

SCENARIOS = {"none": (0, 0), "azi": (3, 0), "ele": (0, 5), "both": (3, 5)}


@pytest.fixture(scope="module")
def eng256(jf, hrir):
    e = jf.Engine(256, 512, 1, hrir=hrir, max_batch_blocks=16)
    yield e
    e.close()


def test_table_matches_oracle(eng256, hrir):
    """a4: device HRTF spectra vs float64 rfft; the reference's own table check uses 1e-6
    (precision_test.cu:209)."""
    t = eng256.read_table()
    ref = model64.build_table(hrir, 1024)
    assert t.shape == ref.shape
    err = np.abs(t - ref).max()
    assert err <= 1e-6, err
    assert np.all(t[:, :, 0].imag == 0) and np.all(t[:, :, 512].imag == 0)


def test_rfft_kernel(eng256):
    """a6: the LDS Stockham FFT against numpy float64."""
    rng = np.random.default_rng(1)
    w = rng.uniform(-0.5, 0.5, (16, 1024)).astype(np.float32)
    w[0] = 0
    w[1] = 0
    w[1, 0] = 1.0           # impulse
    w[2] = 0
    w[2, 1023] = 1.0
    w[3] = 1.0              # DC
    w[4] = np.cos(np.pi * np.arange(1024))  # Nyquist
    sp = eng256.rfft_device(w)
    ref = np.fft.rfft(w.astype(np.float64), axis=-1)
    # relative to the spectrum's scale (sqrt(N) * rms for noise)
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1.0)
    assert (np.abs(sp - ref) / scale).max() <= 5e-7


def test_interp_kernel_exhaustive(eng256):
    """a2/a3: indices and weights bit-exact against the C oracle for every integer
    (ele, azi) the setters can produce, plus out-of-range azimuths and elevations."""
    eles, azis = np.meshgrid(np.arange(-52, 94), np.arange(-10, 372), indexing="ij")
    eles = eles.reshape(-1).astype(np.float32)
    azis = azis.reshape(-1).astype(np.float32)
    rows, w, nt = eng256.interp_device(eles, azis)
    bad = 0
    for i in range(len(eles)):
        r = oracle_lib.interp(float(eles[i]), float(azis[i]))
        if r is None:
            bad += nt[i] != 0
            continue
        orows, ow = oracle_lib.terms(*r)
        n = len(orows)
        ok = nt[i] == n and np.array_equal(rows[i, :n], orows) and np.array_equal(w[i, :n], ow)
        bad += not ok
    assert bad == 0


def test_interp_kernel_fractional_positions(eng256, jf):
    """The same for positions a latched record may carry but no setter produces: fractional degrees over the whole range and
    densely round the ends of the range, where the reference's truncating statements decide what is still a position --
    (-50, 91): an elevation of 90.x lies on the 90-degree ring twice (tests/test_gpu_random_sessions.py found the engine
    refusing it) -- and the host's twin of the rule (jf_interpolation) with them."""
    rng = np.random.default_rng(9)
    eles = np.concatenate([rng.uniform(-52, 93, 20000), rng.uniform(89.5, 91.5, 3000), rng.uniform(-50.5, -48.5, 3000),
                           np.array([90.0, 90.25, 90.999, 91.0, -49.999, -50.0, -40.0, -39.999, 0.0, -0.5, 0.5])]).astype(np.float32)
    azis = np.concatenate([rng.uniform(-5, 365, len(eles) - 11), np.array([0, 359.9, 360, 0.1, 5, 355, 180, 6.43, 6.42, 353.6, 0.5])]).astype(np.float32)
    rows, w, nt = eng256.interp_device(eles, azis)
    bad = n_valid = 0
    for i in range(len(eles)):
        r = oracle_lib.interp(float(eles[i]), float(azis[i]))
        h = jf.interpolation(float(eles[i]), float(azis[i]))
        if r is None:
            bad += (nt[i] != 0) + (h is not None)
            continue
        n_valid += 1
        bad += not (h is not None and np.array_equal(h[0], r[0]) and np.array_equal(h[1], r[1]))
        orows, ow = oracle_lib.terms(*r)
        n = len(orows)
        bad += not (nt[i] == n and np.array_equal(rows[i, :n], orows) and np.array_equal(w[i, :n], ow))
    assert bad == 0 and n_valid > 20000


def test_corrected_rule_kernel_and_end_to_end(jf, hrir, castanets):
    """JF_FLAG_CORRECTED_INTERPOLATION: the kernel's indices/weights bit-exact against the oracle's corrected rule,
    then whole blocks (batch and per-block paths) against the oracle run with the same rule -- including the
    places where the reference's rule misbehaves (negative elevations, azimuth 355..360, the 6.43-degree rings)."""
    F = jf.JF_FLAG_CORRECTED_INTERPOLATION
    eng = jf.Engine(256, 512, 6, hrir=hrir, max_batch_blocks=8, flags=F)
    eles, azis = np.meshgrid(np.arange(-52, 94, 1.5), np.arange(-10, 372, 1.7), indexing="ij")
    eles = eles.reshape(-1).astype(np.float32)
    azis = azis.reshape(-1).astype(np.float32)
    rows, w, nt = eng.interp_device(eles, azis)
    bad = 0
    for i in range(len(eles)):
        r = oracle_lib.interp(float(eles[i]), float(azis[i]), corrected=True)
        if r is None:
            bad += nt[i] != 0
            continue
        orows, ow = oracle_lib.terms(*r)
        n = len(orows)
        bad += not (nt[i] == n and np.array_equal(rows[i, :n], orows) and np.array_equal(w[i, :n], ow))
    assert bad == 0
    ora = oracle_lib.Engine(256, 512, 6, hrir)
    ora.set_mode(2)
    ref = jf.Engine(256, 512, 6, hrir=hrir, max_batch_blocks=8)        # the reference's rule
    spots = [(-5, 10), (0, 358), (45, 10), (-35, 200), (85, 20), (20, 40)]
    for s in range(6):
        for x in (eng, ora, ref):
            x.set_signal(s, 0.5 * castanets[2000 * s: 2000 * s + 7000])
    pos = np.zeros((8, 6, 5), np.float32)
    for k in range(8):
        for s, (e, a) in enumerate(spots):
            pos[k, s] = jf.position_from_spherical(e, (a + 3 * k) % 360, 0.6)
    got, want, other = eng.process_batch(pos), ora.process_batch(pos), ref.process_batch(pos)
    assert np.abs(want).max() > 0.05
    assert np.abs(got - want).max() <= TOL32 * 2
    assert np.abs(got - other).max() > 1e-3        # the flag really changes the rendering at these positions
    for k in range(3):                              # per-block (real-time kernel) path
        for s, (e, a) in enumerate(spots):
            for x in (eng, ora):
                x.set_spherical(s, e, (a + 7 * k) % 360, 0.6)
        assert np.abs(eng.process_block() - ora.process_block()).max() <= TOL32 * 2
    for x in (eng, ref):
        x.close()
    with pytest.raises(jf.JfError):
        jf.Engine(256, 512, 1, hrir=hrir, flags=8)


@pytest.mark.parametrize("B", [256, 128])
@pytest.mark.parametrize("name", list(SCENARIOS))
def test_scenarios_vs_golden(jf, hrir, castanets, golden, B, name):
    """The four benchmarkTesting scenarios (short form) against the committed float64
    vectors and the float32 C oracle, block by block through jf_process_block."""
    azi0, ele0 = SCENARIOS[name]
    ref = golden[f"B{B}_{name}"]
    eng = jf.Engine(B, 512, 1, hrir=hrir)
    ora = oracle_lib.Engine(B, 512, 1, hrir)
    for e in (eng, ora):
        e.set_signal(0, castanets)
        e.reset(0)
    out, out32 = [], []
    for (ele, azi, r) in scenario_positions(azi0, ele0, 3, 3):
        eng.set_spherical(0, ele, azi, r)
        ora.set_spherical(0, ele, azi, r)
        out.append(eng.process_block())
        out32.append(ora.process_block())
    out, out32 = np.array(out), np.array(out32)
    eng.close()
    assert np.abs(ref).max() > 0.05  # the excerpt is not silence
    assert np.abs(out - ref).max() <= TOL64
    assert np.abs(out - out32).max() <= TOL32


def test_batch_equals_blockwise(jf, hrir, castanets):
    """jf_process_batch over K blocks == K calls of jf_process_block, bit for bit
    (same kernels, state carried in HBM), for several sources with different
    trajectories, including a batch boundary in the middle of a crossfade run."""
    S, K, B = 5, 11, 256
    rng = np.random.default_rng(7)
    pos = np.zeros((K, S, 5), np.float32)
    for s in range(S):
        ele, azi = int(rng.integers(-40, 90)), int(rng.integers(0, 360))
        for b in range(K):
            if b % 3 == s % 3:
                azi = (azi + 7) % 360
            pos[b, s] = jf.position_from_spherical(ele, azi, 0.5 + 0.3 * s)
    e1 = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=4)
    e2 = jf.Engine(B, 512, S, hrir=hrir)
    for s in range(S):
        sig = np.roll(castanets, 1000 * s)[: 3000 + 517 * s]  # short, odd lengths -> wraps
        e1.set_signal(s, sig)
        e2.set_signal(s, sig)
    mix1 = e1.process_batch(pos)
    mix2 = []
    for b in range(K):
        for s in range(S):
            p = pos[b, s]
            # latched records are (ele, azi, x, y, z); feed the same through the Cartesian-free path
            e2.set_spherical(s, p[0], p[1], 0.5 + 0.3 * s)
        mix2.append(e2.process_block())
    mix2 = np.array(mix2)
    e1.close()
    e2.close()
    assert np.array_equal(mix1, mix2)


def test_multi_source_mix_vs_oracle(jf, hrir, castanets):
    """a12: several moving sources, all four interpolation cases, mix against both oracles."""
    S, K, B = 6, 8, 256
    starts = [(0, 0), (0, 3), (5, 0), (5, 3), (-15, 7), (85, 20)]
    pos = np.zeros((K, S, 5), np.float32)
    for s, (ele, azi) in enumerate(starts):
        for b in range(K):
            a = (azi + (b // 2) * (s + 1)) % 360
            pos[b, s] = jf.position_from_spherical(ele, a, 0.4 + 0.5 * s)
    eng = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=K)
    ora = oracle_lib.Engine(B, 512, S, hrir)
    mod = model64.Model(B, 512, S, hrir)
    for s in range(S):
        sig = 0.2 * np.roll(castanets, 4321 * s)[:20000]
        for e in (eng, ora, mod):
            e.set_signal(s, sig)
    mix = eng.process_batch(pos)
    mix32 = ora.process_batch(pos)
    mix64, _ = mod.process_batch(pos)
    eng.close()
    assert np.abs(mix - mix64).max() <= TOL64 * 2  # six sources summed
    assert np.abs(mix - mix32).max() <= TOL32 * 2


@pytest.mark.parametrize("B", [128, 256])
def test_full_benchmark_testing_harness(jf, hrir, castanets, B):
    """BASELINE.json configs[0]/[1]: the reference's complete end-to-end harness, not a short form:
    benchmarkTesting (precision_test.cu:2154-2201) = 4 start positions x (172 blocks per position,
    azimuth + 5 degrees x 72 rounds) = 4 x 12 556 blocks, default input looped, r = 0.5.  The HIP
    path (batched through the C ABI) against the float32 C oracle over EVERY block, and against the
    float64 model over the first 3 positions of each scenario."""
    n_dwell, n_rounds = 172, 72
    worst32 = worst64 = 0.0
    for name, (azi0, ele0) in SCENARIOS.items():
        traj = scenario_positions(azi0, ele0, n_dwell, n_rounds)
        pos = np.stack([jf.position_from_spherical(e, a, r) for (e, a, r) in traj])[:, None, :]
        eng = jf.Engine(B, 512, 1, hrir=hrir, max_batch_blocks=512)
        ora = oracle_lib.Engine(B, 512, 1, hrir)
        for x in (eng, ora):
            x.set_signal(0, castanets)
            x.reset(0)
        got = eng.process_batch(pos)
        want = ora.process_batch(pos, n_threads=1)
        eng.close()
        assert got.shape == (n_dwell * (n_rounds + 1), 2 * B)
        worst32 = max(worst32, float(np.abs(got - want).max()))
        mod = model64.Model(B, 512, 1, hrir)
        mod.set_signal(0, castanets)
        mod.reset(0)
        m64, _ = mod.process_batch(pos[: 3 * n_dwell])
        worst64 = max(worst64, float(np.abs(got[: 3 * n_dwell] - m64).max()))
    print(f"B={B}: max|hip-oracle32| = {worst32:.3e}, max|hip-model64| = {worst64:.3e}")
    assert worst32 <= TOL32
    assert worst64 <= TOL64
