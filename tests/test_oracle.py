"""CPU tests of the oracle (runs without a GPU): the C restatement, the float64 model and
the committed fixtures must tell one story, and everything the reference itself pins
(ring sizes, HRIR files, test scenarios, tolerances) must hold."""
import json
import os

import numpy as np
import pytest

import model64
import oracle_lib
from conftest import GOLD, scenario_positions

# the reference's own CPU-vs-GPU tolerance (precision_test.cu:2158, Precision_Check.py:12)
REF_TOL = 2e-7

SCENARIOS = {"none": (0, 0), "azi": (3, 0), "ele": (0, 5), "both": (3, 5)}


def test_ring_sizes_pinned_by_reference_comment():
    """hrtf_signals.cu:10: 56+60+72+72+72+72+72+60+56+45+36+24+12+1 = 710 (= NUM_HRTF)."""
    sizes = [56, 60, 72, 72, 72, 72, 72, 60, 56, 45, 36, 24, 12, 1]
    assert sum(sizes) == 710
    off = oracle_lib.azimuth_offsets()
    assert [off[i + 1] - off[i] for i in range(14)] == sizes
    assert off == model64.AZIMUTH_OFFSET


def test_table_positions_and_fixture_layout(hrir):
    pos = oracle_lib.table_positions()
    assert pos == model64.table_positions()
    fx = np.load(os.path.join(GOLD, "kemar_positions_710x2_i16.npy"))
    assert [tuple(p) for p in fx.tolist()] == pos
    assert hrir.shape == (710, 2, 128)
    # mirror convention: row (e, a) and row (e, 360 - a) carry exchanged ears
    index = {p: j for j, p in enumerate(pos)}
    checked = 0
    for (e, a), j in index.items():
        if 0 < a < 180 and (e, 360 - a) in index:
            k = index[(e, 360 - a)]
            assert np.array_equal(hrir[j, 0], hrir[k, 1]) and np.array_equal(hrir[j, 1], hrir[k, 0])
            checked += 1
    assert checked > 300
    # source on the right (azimuth 90): right ear louder and earlier
    j = index[(0, 90)]
    assert (hrir[j, 1] ** 2).sum() > 10 * (hrir[j, 0] ** 2).sum()


def test_interp_known_answers():
    """SURVEY.md Appendix B (derived from SoundSource.cu:65-105 independently of this repo's code)."""
    appendix_b = {
        (0, 0): ([260, 260, 260, 260], 1), (0, 3): ([260, 261, 260, 261], 2),
        (5, 0): ([260, 260, 332, 332], 3), (5, 3): ([260, 261, 332, 333], 4),
        (10, 5): ([333] * 4, 1), (0, 8): ([261, 262, 261, 262], 2),
        (5, 15): ([263, 263, 335, 335], 3), (-5, 10): ([262] * 4, 1),
        (3, 23): ([264, 265, 336, 337], 4), (8, 18): ([263, 264, 335, 336], 4),
        (0, 358): ([331] * 4, 1), (-15, 7): ([189, 190, 261, 262], 4),
        (45, 10): ([537, 538, 593, 594], 4), (85, 20): ([697, 698, 709, 709], 4),
    }
    for (ele, azi), (idx, case) in appendix_b.items():
        got, _ = oracle_lib.interp(ele, azi)
        assert got.tolist() == idx, (ele, azi)
        assert oracle_lib.lib().jfo_case(oracle_lib.iptr(got)) == case
    _, om = oracle_lib.interp(-15, 7)
    assert np.allclose(om, [.4, .6, .4, .6, -.5, 1.5])
    _, om = oracle_lib.interp(45, 10)
    assert np.allclose(om[:2], [.6221, .3110], atol=1e-4)  # weights that do not sum to 1
    assert oracle_lib.pick_hrtf(0, 0) == 260 and oracle_lib.pick_hrtf(0, 270) == 314

    known = json.load(open(os.path.join(GOLD, "interp_known.json")))
    for key, v in known["points"].items():
        ele, azi = (int(t) for t in key.split(","))
        idx, om = oracle_lib.interp(ele, azi)
        assert idx.tolist() == v["idx"] and om.tolist() == v["omegas"]


def test_interp_c_vs_numpy_exhaustive():
    """Two independent restatements agree bit for bit on every latched (ele, azi)."""
    for ele in range(-52, 94, 1):
        for azi in list(range(0, 361, 7)) + [359, 360]:
            a, b = oracle_lib.interp(ele, azi), model64.interp(ele, azi)
            assert (a is None) == (b is None)
            if a is not None:
                assert a[0].tolist() == b[0] and a[1].tolist() == [float(x) for x in b[1]]
    for azi in range(0, 361):
        a, b = oracle_lib.interp(37, azi), model64.interp(37, azi)
        assert a[0].tolist() == b[0] and a[1].tolist() == [float(x) for x in b[1]]


def test_corrected_rule_restatements_and_properties():
    """JF_FLAG_CORRECTED_INTERPOLATION (the corrected variant behind a flag; not in the
    reference): C oracle == NumPy model bit for bit, and the properties the reference's rule lacks."""
    import itertools
    for ele, azi in itertools.product(np.arange(-45, 91, 1.5), np.arange(-10, 371, 2.3)):
        a, b = oracle_lib.interp(float(ele), float(azi), corrected=True), model64.interp_corrected(ele, azi)
        assert (a is None) == (b is None)
        assert a[0].tolist() == b[0] and a[1].tolist() == [float(x) for x in b[1]]
        om = a[1]
        assert (om >= 0).all() and (om <= 1).all()                       # no extrapolation
        assert om[0] + om[1] == 1 and om[2] + om[3] == 1 and om[4] + om[5] == 1
    # negative elevations: true floor (the reference gives rings (0, 0) and a weight of -0.5 here)
    idx, om = oracle_lib.interp(-5, 10, corrected=True)
    assert idx.tolist() == [190, 190, 262, 262] and om[4] == 0.5
    assert oracle_lib.interp(-5, 10)[0].tolist() == [262] * 4 and oracle_lib.interp(-5, 10)[1][4] == -0.5
    # wrap from a ring's last azimuth to its first (the reference picks 355 degrees twice at azimuth 358)
    idx, om = oracle_lib.interp(0, 358, corrected=True)
    assert idx.tolist() == [331, 260, 331, 260] and abs(om[0] - 0.6) < 1e-6
    assert oracle_lib.interp(0, 358)[0].tolist() == [331] * 4
    # 6.43-degree ring: weights sum to 1 (reference: 0.6221 + 0.3110)
    idx, om = oracle_lib.interp(45, 10, corrected=True)
    assert idx.tolist() == [537, 538, 593, 594] and om[0] + om[1] == 1
    # below the lowest ring: clamped to it; above 90 or non-finite: invalid
    assert oracle_lib.interp(-47, 30, corrected=True)[0].tolist() == oracle_lib.interp(-40, 30, corrected=True)[0].tolist()
    assert oracle_lib.interp(91, 0, corrected=True) is None and oracle_lib.interp(0, float("nan"), corrected=True) is None
    # where the reference's rule is sound (ele >= 0 on a 5-degree ring, azimuth < 355) the two rules agree
    for ele, azi in ((5, 3), (0, 0), (10, 5), (3, 23), (8, 18)):
        r, c = oracle_lib.interp(ele, azi), oracle_lib.interp(ele, azi, corrected=True)
        rt, ct = oracle_lib.terms(*r), oracle_lib.terms(*c)
        assert rt[0].tolist() == ct[0].tolist() and np.abs(rt[1] - ct[1]).max() < 1e-6


def test_geometry():
    """SoundSource.cu:20-54 incl. the handedness mismatch of the two setters (App. C#15,16)."""
    for ele, azi, r in [(0, 0, .5), (0, 90, 1), (30, 45, 2), (-40, 359, 3.5), (5.4, 2.6, .5)]:
        c = oracle_lib.from_spherical(ele, azi, r)
        e2, a2, xyz = model64.from_spherical(ele, azi, r)
        assert c.tolist() == [e2, a2, xyz[0], xyz[1], xyz[2]]
    c = oracle_lib.from_spherical(0, 90, 1.0)
    assert c[2] == pytest.approx(1.0) and abs(c[4]) < 1e-6   # +x
    back = oracle_lib.from_cartesian(c[2], c[3], c[4])
    assert back[1] == 270.0                                   # Cartesian maps +x to 270
    assert oracle_lib.from_cartesian(0, 0, 0) is None
    for xyz in [(0.3, 0.1, -0.4), (-1, 0.5, 2), (0, 0, .5), (0, 1, 0)]:
        c = oracle_lib.from_cartesian(*xyz)
        m = model64.from_cartesian(*xyz)
        assert c.tolist() == [float(v) for v in m]


def test_distance_factor():
    d = oracle_lib.distance_factor(0.0, 0.0, 0.5, 513)
    ref = model64.distance_factor((0.0, 0.0, 0.5), 513)
    assert np.abs(d - ref).max() < 6e-8
    r, fsvs, frac = model64.distance_params((0.0, 0.0, 0.5))
    assert d[0].real == np.float32(1.0 / float(frac)) and d[0].imag == 0.0
    # gain 1/(1 + fsvs r'^2) with r' = |coords| / 5
    assert float(frac) == pytest.approx(1 + 44100 / 343 * 0.01, rel=1e-6)


def test_oracle_fft_against_numpy():
    rng = np.random.default_rng(0)
    x = rng.uniform(-.5, .5, 1024).astype(np.float32)
    X = np.fft.rfft(x.astype(np.float64))
    assert np.abs(oracle_lib.rfft(x) - X).max() / np.abs(X).max() < 5e-7
    Xc = X.astype(np.complex64)
    y = np.fft.irfft(Xc.astype(np.complex128), 1024) * 1024
    assert np.abs(oracle_lib.irfft(Xc, 1024) - y).max() / np.abs(y).max() < 5e-7
    # c2r ignores the imaginary parts of bins 0 and N/2
    Xd = Xc.copy()
    Xd[0] += 3j
    Xd[512] -= 2j
    assert np.array_equal(oracle_lib.irfft(Xd, 1024), oracle_lib.irfft(Xc, 1024))


def test_table_oracle_vs_model(hrir):
    t32 = oracle_lib.build_table(hrir, 1024)
    t64 = model64.build_table(hrir, 1024)
    assert np.abs(t32 - t64).max() <= 1e-6  # the reference's own table tolerance (precision_test.cu:209)


@pytest.mark.parametrize("B", [256, 128])
@pytest.mark.parametrize("name", list(SCENARIOS))
def test_scenarios_oracle_vs_golden(hrir, castanets, golden, B, name):
    """C oracle vs the committed float64 vectors, within the reference's 2e-7."""
    azi0, ele0 = SCENARIOS[name]
    ref = golden[f"B{B}_{name}"]
    ora = oracle_lib.Engine(B, 512, 1, hrir)
    ora.set_signal(0, castanets)
    ora.reset(0)
    out = []
    for (ele, azi, r) in scenario_positions(azi0, ele0, 3, 3):
        ora.set_spherical(0, ele, azi, r)
        out.append(ora.process_block())
    assert np.abs(np.array(out) - ref).max() <= REF_TOL


def test_golden_matches_model(hrir, castanets, golden):
    """The committed vectors are reproducible from oracle/model64.py on any machine."""
    mod = model64.Model(256, 512, 1, hrir)
    mod.set_signal(0, castanets)
    mod.reset(0)
    out = []
    for (ele, azi, r) in scenario_positions(3, 5, 3, 3):
        mod.set_spherical(0, ele, azi, r)
        out.append(mod.process_block())
    assert np.abs(np.array(out) - golden["B256_both"]).max() < 1e-12


def test_first_block_crossfades_from_origin(hrir, castanets):
    """App. A: old position starts at (0, 0), so a first latched position elsewhere
    cross-fades from the front HRTF; staying put does not."""
    a = oracle_lib.Engine(256, 512, 1, hrir)
    b = oracle_lib.Engine(256, 512, 1, hrir)
    for e in (a, b):
        e.set_signal(0, castanets[5000:])
    a.set_spherical(0, 0, 90, 0.5)
    b.set_spherical(0, 0, 90, 0.5)
    ya0 = a.process_block()
    b.reset(0)
    yb0 = b.process_block()
    assert np.array_equal(ya0, yb0)
    ya1, yb1 = a.process_block(), b.process_block()
    assert np.array_equal(ya1, yb1)
    # frame 0 of the faded block equals the frame computed with the OLD (front) filter only
    c = oracle_lib.Engine(256, 512, 1, hrir)
    c.set_signal(0, castanets[5000:])
    c.set_spherical(0, 0, 0, 0.5)
    # same distance -> same D; front filter
    yc0 = c.process_block()
    assert np.allclose(ya0[:2], yc0[:2], atol=1e-7)


def test_wrap_and_short_signals(hrir):
    """Audio.cu:121-139: looped feed; lengths around B, not multiples of B, and empty."""
    rng = np.random.default_rng(3)
    for n in (0, 1, 100, 255, 256, 257, 1000):
        sig = rng.uniform(-.5, .5, n).astype(np.float32)
        ora = oracle_lib.Engine(256, 512, 1, hrir)
        mod = model64.Model(256, 512, 1, hrir)
        for e in (ora, mod):
            e.set_signal(0, sig)
            e.set_spherical(0, 5, 3, 0.5)
        for _ in range(6):
            y32, y64 = ora.process_block(), mod.process_block()
            assert np.abs(y32 - y64).max() <= REF_TOL
        if n == 0:
            assert not y32.any()


def test_batch_equals_blockwise_and_mix_order(hrir, castanets):
    S, K = 3, 7
    pos = np.zeros((K, S, 5), np.float32)
    for s in range(S):
        for b in range(K):
            pos[b, s] = oracle_lib.from_spherical(10 * s - 5, (13 * s + 4 * (b // 2)) % 360, 0.5 + s)
    e1 = oracle_lib.Engine(128, 512, S, hrir)
    e2 = oracle_lib.Engine(128, 512, S, hrir)
    for s in range(S):
        e1.set_signal(s, castanets[1000 * s: 1000 * s + 5000])
        e2.set_signal(s, castanets[1000 * s: 1000 * s + 5000])
    mix, part = e1.process_batch(pos, want_partial=True, n_threads=2)
    blockwise = []
    for b in range(K):
        for s in range(S):
            e2.set_spherical(s, pos[b, s, 0], pos[b, s, 1], 0.5 + s)
        blockwise.append(e2.process_block())
    assert np.array_equal(mix, np.array(blockwise))
    serial = np.zeros_like(mix)
    for s in range(S):
        serial = serial + part[s]
    assert np.array_equal(mix, serial)  # Audio.cu:109-110 order


def test_linearity_and_gain(hrir, castanets):
    """Domain properties used at full size on the GPU: the path is linear in the signal."""
    sig = castanets[:8192]
    outs = []
    for g in (1.0, 0.5):
        e = oracle_lib.Engine(256, 512, 1, hrir)
        e.set_signal(0, (g * sig).astype(np.float32))
        e.set_spherical(0, 20, 123, 1.0)
        outs.append(np.array([e.process_block() for _ in range(4)]))
    assert np.abs(outs[0] * 0.5 - outs[1]).max() < 1e-7


# ------------------------------------------------------------ convolution reverb --
def _ir(n, seed=99, decay=4.0):
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(n) * np.exp(-decay * np.arange(n) / n)
    return (h / np.sqrt((h ** 2).sum())).astype(np.float32)


def test_reverb_offline_is_the_reference_whole_signal_form(castanets):
    """jfo_reverb_offline against float64: circular convolution of length new_size = n + (n_ir - n_ir/2)
    (PadData, kernels.cu:169-188; cudaPart.cu:87-153) times rms / rms2 (cudaPart.cu:118,161-165)."""
    for n, n_ir in [(6000, 3001), (5000, 700), (2048, 2048)]:
        x = castanets[1000:1000 + n]
        h = _ir(n_ir)
        got, g = oracle_lib.reverb_offline(x, h)
        new_size = n + (n_ir - n_ir // 2)
        assert len(got) == new_size
        X = np.fft.rfft(np.pad(x.astype(np.float64), (0, new_size - n)))
        H = np.fft.rfft(np.pad(h.astype(np.float64), (0, new_size - n_ir)))
        y = np.fft.irfft(X * H, new_size)
        want_g = np.sqrt((x.astype(np.float64) ** 2).sum() / (y ** 2).sum())
        assert g == pytest.approx(want_g, rel=2e-6)
        want = want_g * y
        assert np.abs(want).max() > 0.05
        # float32 transforms of 16 k points: ~log2(m) roundings of the largest sample
        assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()


@pytest.mark.parametrize("B,n_ir", [(128, 5 * 128 + 37), (256, 256 * 3), (64, 40), (128, 128 * 60 + 1)])
def test_reverb_stream_form_vs_float64_convolution(hrir, castanets, B, n_ir):
    """The stream form (jfo_reverb_set_ir) ahead of the spatialiser: the C oracle's stereo blocks against the
    float64 model fed gain * (looped dry stream (*) ir) computed by direct float64 convolution.  Tolerance:
    the spatialiser's 4e-7 (float32 C oracle against float64) + 1e-7 sqrt(P) for the float32 sum over P partitions."""
    S, K = 2, 14
    ir = _ir(n_ir)
    gain = 4.0
    P = -(-n_ir // B)
    sigs = [castanets[3000:3000 + 2 * B + 77], castanets[9000:9000 + 5000]]   # the first loops inside a block
    pos = np.zeros((K, S, 5), np.float32)
    for b in range(K):
        for s in range(S):
            pos[b, s] = oracle_lib.from_spherical(10 * s - 5, (33 * s + 4 * (b // 2)) % 360, 0.5 + 0.6 * s)
    ora = oracle_lib.Engine(B, 512, S, hrir)
    mod = model64.Model(B, 512, S, hrir)
    for s in range(S):
        ora.set_signal(s, sigs[s])
        stream = np.tile(sigs[s].astype(np.float64), -(-K * B // len(sigs[s])))[:K * B]
        mod.src[s].buf = gain * np.convolve(stream, ir.astype(np.float64))[:K * B]   # kept in float64
        mod.src[s].count = 0
    ora.set_reverb(ir, gain)
    got, part = ora.process_batch(pos, want_partial=True)
    want, wpart = mod.process_batch(pos)
    ora.close()
    tol = (4e-7 + 1e-7 * np.sqrt(P)) * max(1.0, np.abs(wpart).max())
    assert np.abs(wpart).max() > 0.02
    assert np.abs(part - wpart).max() <= tol
    assert np.abs(got - want).max() <= tol * S


def test_reverb_stream_form_reaches_the_offline_form(hrir, castanets):
    """Ties the two forms together: the stream form run over the zero-padded signal on a loop (PadData's new_size
    samples, what the reference loops as `buf` after cudaFFT) equals, from the second pass on, the spatialiser fed the
    offline result -- the circular wrap of the reference's whole-signal product IS the previous pass's tail."""
    B, n, n_ir = 128, 4 * 128 * 5, 1100
    x = castanets[2000:2000 + n]
    ir = _ir(n_ir)
    buf, g = oracle_lib.reverb_offline(x, ir)
    new_size = len(buf)
    assert new_size == n + 550 and new_size % B != 0
    K = 2 * (-(-new_size // B)) + 3
    pos = np.zeros((K, 1, 5), np.float32)
    for b in range(K):
        pos[b, 0] = oracle_lib.from_spherical(5, (3 + b) % 360, 0.7)
    a = oracle_lib.Engine(B, 512, 1, hrir)
    a.set_signal(0, np.pad(x, (0, new_size - n)))
    a.set_reverb(ir, g)
    b_ = oracle_lib.Engine(B, 512, 1, hrir)
    b_.set_signal(0, buf)
    ya = a.process_batch(pos)
    yb = b_.process_batch(pos)
    a.close()
    b_.close()
    first = -(-new_size // B) + 8     # blocks whose 1024-sample window lies wholly in the second pass
    assert np.abs(yb[first:]).max() > 0.02
    assert np.abs(ya[first:] - yb[first:]).max() <= 2e-6 * max(1.0, np.abs(yb).max())
    assert np.abs(ya[:4] - yb[:4]).max() > 1e-3     # the first pass has no tail wrapped onto it yet
