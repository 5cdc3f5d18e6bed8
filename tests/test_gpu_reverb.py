"""Convolution reverb ahead of the spatialiser (BASELINE.json configs[4]) on
a real MI355X: the partitioned FDL convolution against a float64 full convolution, chained into
the spatialiser model.

Oracle: the wet stream is `gain * (looped dry stream (*) ir)` computed in float64 (np.convolve /
scipy fftconvolve), then oracle/model64.py spatialises it.  Tolerance: the partitioned float32
accumulation over P partitions adds ~sqrt(P) * eps relative error to the wet signal, so the bound
is 2e-7 (the spatialiser's) + 1e-7 * sqrt(P), relative to max(1, |y|) -- stated per test.
"""
import numpy as np
import pytest

import model64

pytestmark = pytest.mark.gpu


def _ir(n, seed=99, decay=4.0):
    """Exponentially decaying noise, unit energy."""
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(n) * np.exp(-decay * np.arange(n) / n)
    return (h / np.sqrt((h ** 2).sum())).astype(np.float32)


def _wet_stream(dry, n_total, ir, gain):
    """float64: gain * (dry looped to n_total samples, zero before the start) convolved with ir."""
    from scipy.signal import fftconvolve
    reps = -(-n_total // len(dry))
    stream = np.tile(dry.astype(np.float64), reps)[:n_total]
    if len(ir) * n_total < 5e7:
        wet = np.convolve(stream, ir.astype(np.float64))[:n_total]
    else:
        wet = fftconvolve(stream, ir.astype(np.float64))[:n_total]
    return gain * wet


def _run(jf, hrir, B, S, K, max_k, ir, gain, sigs, pos, blockwise=False, form=0):
    eng = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=max_k)
    eng.set_reverb_form(form)
    for s in range(S):
        eng.set_signal(s, sigs[s])
    eng.set_reverb(ir, gain)
    if blockwise:
        out = []
        for b in range(K):
            for s in range(S):
                eng.set_spherical(s, pos[b, s, 0], pos[b, s, 1], 0.5 + 0.4 * s)
            out.append(eng.process_block())
        mix = np.array(out)
    else:
        mix = eng.process_batch(pos)
    eng.close()
    return mix


def _model(hrir, B, S, K, ir, gain, sigs, pos):
    mod = model64.Model(B, 512, S, hrir)
    for s in range(S):
        wet = _wet_stream(sigs[s], K * B, ir, gain)
        # the model stores float32 samples; keep the float64 wet stream exactly instead
        mod.src[s].buf = wet
        mod.src[s].count = 0
    mix, _ = mod.process_batch(pos)
    return mix


def _positions(jf, S, K):
    pos = np.zeros((K, S, 5), np.float32)
    for s in range(S):
        for b in range(K):
            pos[b, s] = jf.position_from_spherical(-20 + 25 * s, (40 * s + 3 * (b // 2)) % 360, 0.5 + 0.4 * s)
    return pos


@pytest.mark.parametrize("B", [128, 256, 64])
def test_reverb_short_ir_vs_float64(jf, hrir, castanets, B):
    S, K = 3, 22
    ir = _ir(5 * B + 37)           # 6 partitions, ragged last one
    gain = 0.7
    sigs = [castanets[4000 * s: 4000 * s + 9000 + 123 * s] for s in range(S)]   # loop inside the run
    pos = _positions(jf, S, K)
    got = _run(jf, hrir, B, S, K, 5, ir, gain, sigs, pos)
    want = _model(hrir, B, S, K, ir, gain, sigs, pos)
    tol = (2e-7 + 1e-7 * np.sqrt(6)) * max(1.0, np.abs(want).max()) * S
    assert np.abs(want).max() > 0.05
    assert np.abs(got - want).max() <= tol


def test_reverb_two_second_ir(jf, hrir, castanets):
    """configs[4] geometry: B = 128, 2.0 s IR = 88 200 taps = 690 partitions of 128."""
    B, S, K = 128, 2, 12
    ir = _ir(88200)
    assert -(-len(ir) // B) == 690
    sigs = [castanets[:30000], castanets[30000:52000]]
    pos = _positions(jf, S, K)
    got = _run(jf, hrir, B, S, K, 4, ir, 1.0, sigs, pos)
    want = _model(hrir, B, S, K, ir, 1.0, sigs, pos)
    tol = (2e-7 + 1e-7 * np.sqrt(690)) * max(1.0, np.abs(want).max()) * S
    assert np.abs(want).max() > 0.01
    assert np.abs(got - want).max() <= tol


@pytest.mark.parametrize("B,n_ir", [(128, 700), (128, 128 * 83 + 5), (256, 256 * 19), (64, 64 * 3)])
def test_reverb_blockwise_equals_batch(jf, hrir, castanets, B, n_ir):
    """Per-block calls (stage A fused into the multiply-accumulate kernel: the last wave transforms the new block and
    takes partition 0, the other 15 share the rest) against batch calls of the same form with its two kernels:
    bit-identical when that form is pinned for both, and the fused form within the float64 tolerance of the model and
    of the batch.  IR lengths from 3 to 83 partitions: fewer partitions than waves, and many."""
    S, K = 2, 9
    ir = _ir(n_ir)
    P = -(-n_ir // B)
    sigs = [castanets[:6000], castanets[7000:12000]]
    pos = _positions(jf, S, K)
    a = _run(jf, hrir, B, S, K, 4, ir, 0.5, sigs, pos, form=1)
    b = _run(jf, hrir, B, S, K, 1, ir, 0.5, sigs, pos, blockwise=True, form=1)
    assert np.array_equal(a, b)
    c = _run(jf, hrir, B, S, K, 1, ir, 0.5, sigs, pos, blockwise=True)   # default: the fused form
    want = _model(hrir, B, S, K, ir, 0.5, sigs, pos)
    tol = (2e-7 + 1e-7 * np.sqrt(P)) * max(1.0, np.abs(want).max()) * S
    assert np.abs(want).max() > 0.01
    assert np.abs(c - want).max() <= tol
    assert np.abs(c - a).max() <= 2 * tol
    assert P <= 8 or not np.array_equal(c, a)   # really another association


@pytest.mark.parametrize("B,P,max_k", [(128, 21, 11), (256, 9, 7), (64, 70, 16), (128, 3, 8)])
def test_reverb_mac_forms_agree(jf, hrir, castanets, B, P, max_k):
    """The three forms of the multiply-accumulate stage (per (block, source); source groups sharing the IR
    spectra; block tiles sharing a sliding window of input spectra) add the same products in different
    associations: each within the float64 tolerance, and within 2 * tol of one another.  max_k is
    chosen so that calls end in partial tiles (K % 8 != 0), P so that the waves' partition chunks are
    ragged or empty."""
    S, K = 4, 27
    ir = _ir(P * B - 5)
    gain = 0.6
    sigs = [castanets[3000 * s: 3000 * s + 8000 + 77 * s] for s in range(S)]
    pos = _positions(jf, S, K)
    want = _model(hrir, B, S, K, ir, gain, sigs, pos)
    tol = (2e-7 + 1e-7 * np.sqrt(P)) * max(1.0, np.abs(want).max()) * S
    outs = [_run(jf, hrir, B, S, K, max_k, ir, gain, sigs, pos, form=f) for f in (1, 2, 3)]
    assert np.abs(want).max() > 0.05
    for o in outs:
        assert np.abs(o - want).max() <= tol
    assert not np.array_equal(outs[0], outs[2]) or P <= 8   # really different code paths


def test_reverb_identity_ir_is_the_dry_path(jf, hrir, castanets):
    """ir = [1]: the wet signal is the dry signal (to FFT rounding), so the output matches the
    plain engine within the float32 tolerance."""
    B, S, K = 256, 1, 6
    sigs = [castanets[2000:20000]]
    pos = _positions(jf, S, K)
    wet = _run(jf, hrir, B, S, K, 3, np.array([1.0], np.float32), 1.0, sigs, pos)
    dry = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=3)
    dry.set_signal(0, sigs[0])
    ref = dry.process_batch(pos)
    dry.close()
    assert np.abs(wet - ref).max() <= 4e-7


def test_reverb_off_and_unsupported_block(jf, hrir, castanets):
    e = jf.Engine(192, 512, 1, hrir=hrir)
    with pytest.raises(jf.JfError) as ei:
        e.set_reverb(_ir(100))
    assert ei.value.code == jf.JF_ERR_ARG
    e.close()
    # switching the stage off returns to the dry path of a fresh engine
    a = jf.Engine(128, 512, 1, hrir=hrir)
    b = jf.Engine(128, 512, 1, hrir=hrir)
    for x in (a, b):
        x.set_signal(0, castanets[:5000])
        x.set_spherical(0, 10, 200, 1.0)
    a.set_reverb(_ir(300), 0.5)
    for _ in range(3):
        a.process_block()
    a.set_reverb(np.zeros(0, np.float32))
    a.set_signal(0, castanets[:5000])
    for _ in range(4):
        assert np.array_equal(a.process_block(), b.process_block())
    a.close()
    b.close()
