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
import oracle_lib

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


def _run(jf, hrir, B, S, K, max_k, ir, gain, sigs, pos, blockwise=False, form=0, part=0, head_fused=False):
    eng = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=max_k)
    eng.set_reverb_head_fused(head_fused)
    eng.set_reverb_form(form)
    eng.set_reverb_partitioning(part)
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


@pytest.mark.parametrize("B,n_ir", [(128, 16 * 128 * 4 + 77), (256, 8 * 256 * 3 + 5), (64, 16 * 64 * 5), (128, 128 * 40 - 3), (256, 300)])
def test_head_inside_the_realtime_kernel(jf, hrir, castanets, B, n_ir):
    """One-block calls with the reverb's head run by the spatialiser's own waves (rt_block_kernel<.., reverb>: one launch per
    audio block; jf_debug_set_reverb_head_fused) against the float64 model, against the head as a kernel of its own, and with
    the two forms taking turns in ONE stream of blocks (every ring, counter and window they leave is the other's input): non-
    uniformly partitioned responses (head of 2 M partitions, the big partitions on the side stream) and short uniform ones; 11
    sources = two workgroups, the second with three waves; a reset and a new signal on the way."""
    S, K = 11, 70
    ir = _ir(n_ir, decay=3.0)
    gain = 0.6
    sigs = [castanets[3000 * s: 3000 * s + 9000 + 131 * s] for s in range(S)]
    pos = _positions(jf, S, K)
    pos[:, 3:] = pos[:, 3 % 3: 3 % 3 + 1]        # (the helper's radii are per source: keep them inside its 0.5 + 0.4 s range)
    for s_ in range(3, S):
        for b in range(K):
            pos[b, s_] = jf.position_from_spherical(-30 + 9 * s_, (25 * s_ + 2 * b) % 360, 0.6 + 0.1 * s_)
    want = _model(hrir, B, S, K, ir, gain, sigs, pos)
    P = -(-n_ir // B)
    tol = (2e-7 + 1e-7 * np.sqrt(P)) * max(1.0, np.abs(want).max()) * np.sqrt(S) * 2
    outs = {}
    for mode in ("fused", "own", "turns"):
        e = jf.Engine(B, 512, S, hrir=hrir)
        e.set_reverb_head_fused(mode != "own")
        for s_ in range(S):
            e.set_signal(s_, sigs[s_])
        e.set_reverb(ir, gain)
        got = []
        for b in range(K):
            if mode == "turns" and b % 5 == 0:
                e.set_reverb_head_fused((b // 5) % 2 == 0)
            e.set_latched(pos[b])
            got.append(e.process_block())
            want_rt = "rt_block_kernel<%d,8%s>" % (B // 64, ",reverb" if (mode == "fused" or (mode == "turns" and (b // 5) % 2 == 0)) else "")
            assert e.last_kernels()[-1] == want_rt or e.last_kernels()[-1].startswith("reverb_big_fft"), (b, e.last_kernels())
        outs[mode] = np.array(got)
        e.close()
    assert np.abs(want).max() > 0.02
    for mode, got in outs.items():
        assert np.abs(got - want).max() <= tol, mode
    assert not np.array_equal(outs["fused"], outs["own"])


# ------------------------------------------------------------------------------- non-uniform partitioning --
@pytest.mark.parametrize("B,n_big,ragged", [(128, 3, 0), (128, 7, 901), (64, 5, 17), (256, 3, 1000)])
def test_nonuniform_partitioning_against_float64_and_the_uniform_form(jf, hrir, castanets, B, n_big, ragged):
    """A head of 2 M = 32 (B = 256: 16) partitions of B + partitions of M B behind it (jf_debug_set_reverb_partitioning; the default for long
    responses) against gain * float64 convolution -> float64 spatialiser model and against the engine with uniform
    partitions, over 70 blocks = four steps of the big partitions, as ONE run of calls of ragged sizes (1, 5, 16, 17, 31
    blocks: steps at the start, in the middle and at the end of a call, calls without any).  The response is n_big big
    partitions long (+ a ragged rest)."""
    S, K = 3, 70
    M = 16 if B <= 128 else 8        # blocks per big block: big partitions of 1024 or 2048 taps
    B1 = M * B
    n_ir = B1 + n_big * B1 - (B1 - ragged if ragged else 0)
    ir = _ir(n_ir, decay=3.0)
    gain = 0.6
    sigs = [castanets[5000 * s: 5000 * s + 12000 + 91 * s] for s in range(S)]
    pos = _positions(jf, S, K)
    want = _model(hrir, B, S, K, ir, gain, sigs, pos)
    P = -(-n_ir // B)
    tol = (2e-7 + 1e-7 * np.sqrt(P)) * max(1.0, np.abs(want).max()) * S
    outs = {}
    for part in (2, 1):
        for sizes in ((1, 5, 16, 17, 31), (6, 64)):      # 70 blocks either way
            e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=max(sizes))
            e.set_reverb_partitioning(part)
            for s_ in range(S):
                e.set_signal(s_, sigs[s_])
            e.set_reverb(ir, gain)
            n, head, big, taps = e.reverb_partitions()
            assert n == P and (head, big, taps) == ((2 * M, -(-(n_ir - B1) // B1) - 1, B1) if part == 2 else (P, 0, 0))
            got, b0 = [], 0
            for k in sizes:
                got.append(e.process_batch(pos[b0:b0 + k]))
                b0 += k
            assert b0 == K
            outs[part, sizes] = np.concatenate(got)
            e.close()
        outs[part] = outs[part, (1, 5, 16, 17, 31)]
    assert np.abs(want).max() > 0.02
    # (6, 64): blocks in front of the call's first whole big block, whole big blocks, and blocks behind the last, whose
    # big blocks are a multiple of four apart: TAIL of the one must not land on the other's place in the fut ring
    assert np.abs(outs[2, (6, 64)] - want).max() <= tol
    assert np.abs(outs[2] - want).max() <= tol
    assert np.abs(outs[1] - want).max() <= tol
    assert not np.array_equal(outs[1], outs[2])      # really another decomposition of the same convolution
    # the tail matters: a response cut behind the head gives something else
    cut = _model(hrir, B, S, K, ir[:B1], gain, sigs, pos)
    assert np.abs(cut - want).max() > 100 * tol


@pytest.mark.parametrize("B", [128, 256])
@pytest.mark.parametrize("short", ["half", "one", "one_and_a_bit", "two_and_a_bit", "almost_three"])
def test_forced_nonuniform_partitioning_of_short_responses(jf, hrir, castanets, B, short):
    """jf_debug_set_reverb_partitioning(e, 2) with a response the default would never partition non-uniformly (below three
    big partitions): no big partition behind the head at all (the head alone, zero-padded to 2 M partitions), ONE (TAIL
    sums over no spectrum: zeros) and TWO (TAIL is a single product).  Against the float64 convolution, the uniform form,
    and per-block calls (the side stream) against batch calls without a whole big block bit for bit."""
    S, K = 2, 52
    M = 16 if B <= 128 else 8
    B1 = M * B
    n_ir = {"half": B1 // 2 + 3, "one": B1, "one_and_a_bit": B1 + 5, "two_and_a_bit": 2 * B1 + 9, "almost_three": 3 * B1 - 1}[short]
    p1 = {"half": 0, "one": 0, "one_and_a_bit": 1, "two_and_a_bit": 2, "almost_three": 2}[short]
    ir = _ir(n_ir, decay=2.0)
    gain = 0.6
    sigs = [castanets[5000 * s: 5000 * s + 12000 + 91 * s] for s in range(S)]
    pos = _positions(jf, S, K)
    want = _model(hrir, B, S, K, ir, gain, sigs, pos)
    P = -(-n_ir // B)
    tol = (2e-7 + 1e-7 * np.sqrt(P)) * max(1.0, np.abs(want).max()) * S
    outs = {}
    for part, sizes in ((2, (1, 5, 16, 17, 13)), (2, (6, 46)), (2, (7,) * 7 + (3,)), (1, (6, 46))):
        e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=max(sizes))
        e.set_reverb_partitioning(part)
        if sizes[0] == 7:
            e.set_reverb_form(1)       # (pinned like the per-block run below: same sums in the head)
        for s_ in range(S):
            e.set_signal(s_, sigs[s_])
        e.set_reverb(ir, gain)
        n, head, big, taps = e.reverb_partitions()
        assert n == P
        if part == 2:
            assert (head, big, taps) == (2 * M, max(p1 - 1, 0), B1 if p1 else 0), (head, big, taps)
        got, b0 = [], 0
        for k in sizes:
            got.append(e.process_batch(pos[b0:b0 + k]))
            b0 += k
        assert b0 == K
        outs[part, sizes] = np.concatenate(got)
        e.close()
    assert np.abs(want).max() > 0.02
    for key, got in outs.items():
        assert np.abs(got - want).max() <= tol, key
    blockwise = _run(jf, hrir, B, S, K, 1, ir, gain, sigs, pos, blockwise=True, form=1, part=2)
    assert np.array_equal(blockwise, outs[2, (7,) * 7 + (3,)])
    free = _run(jf, hrir, B, S, K, 1, ir, gain, sigs, pos, blockwise=True, part=2)       # the fused head kernel + the side stream
    assert np.abs(free - want).max() <= tol


def test_nonuniform_blockwise_equals_batch_and_the_default_takes_it(jf, hrir, castanets):
    """Per-block calls against batch calls that contain no whole big block (ragged sizes up to 15, boundaries inside and at
    their ends) with the head's form pinned for both: bit-identical -- every block is then head + TAIL(m), and TAIL's products
    are added in the same order whatever the launch.  A batch call of whole big blocks forms their wet signal from the big
    partitions alone (FULL(m): another decomposition): within the float64 tolerance, not bit-identical.  And the 2 s
    response of configs[4] takes the non-uniform form by default: 32 + 42 partitions instead of 690."""
    B, S, K = 128, 2, 80
    ir = _ir(16 * B * 6 + 333, decay=3.0)
    sigs = [castanets[:9000], castanets[10000:17000]]
    pos = _positions(jf, S, K)
    b = _run(jf, hrir, B, S, K, 1, ir, 0.5, sigs, pos, blockwise=True, form=1, part=2)    # 80 calls
    e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=15)
    e.set_reverb_form(1)
    e.set_reverb_partitioning(2)
    for s_ in range(S):
        e.set_signal(s_, sigs[s_])
    e.set_reverb(ir, 0.5)
    got, b0 = [], 0
    for k in (7, 15, 9, 3, 15, 15, 15, 1):
        got.append(e.process_batch(pos[b0:b0 + k]))
        assert not any("reverb_big_mac_kernel<2048,16>" in x for x in e.last_kernels())
        b0 += k
    e.close()
    assert b0 == K and np.array_equal(np.concatenate(got), b)
    a = _run(jf, hrir, B, S, K, 80, ir, 0.5, sigs, pos, form=1, part=2)                   # one call of five whole big blocks
    c = _run(jf, hrir, B, S, K, 1, ir, 0.5, sigs, pos, blockwise=True)                    # default: the fused head kernel in front
    c2 = _run(jf, hrir, B, S, K, 1, ir, 0.5, sigs, pos, blockwise=True, head_fused=True)   # the head inside the real-time kernel
    want = _model(hrir, B, S, K, ir, 0.5, sigs, pos)
    tol = (2e-7 + 1e-7 * np.sqrt(-(-len(ir) // B))) * max(1.0, np.abs(want).max()) * S
    assert np.abs(want).max() > 0.02
    for x in (a, b, c, c2):
        assert np.abs(x - want).max() <= tol
    assert not np.array_equal(a, b) and not np.array_equal(c, c2)
    for fused in (False, True):    # the head as a kernel of its own in front (default), or inside the real-time kernel's launch
        head = [] if fused else ["reverb_mac_kernel<128,1,true>"]
        rt = "rt_block_kernel<2,8,reverb>" if fused else "rt_block_kernel<2,8>"
        e = jf.Engine(128, 512, 1, hrir=hrir)
        e.set_reverb_head_fused(fused)
        e.set_reverb(_ir(88200), 1.0)
        assert e.reverb_partitions() == (690, 32, 42, 2048)
        for _ in range(16):
            e.process_block()
        ks = e.last_kernels()      # the 16th block completes big block 0: X_1, TAIL(2) and its inverse on the side stream
        assert ks == head + ["reverb_big_fft_kernel<2048,1>@side", "reverb_big_mac_kernel<2048,1>@side",
                             "reverb_big_ifft_kernel<2048,1>@side", rt], ks
        e.process_block()          # the 17th block is the first of big block 1: nothing but its head
        assert e.last_kernels() == head + [rt]
        e.set_reverb_async(False)  # from here on everything in line on the engine's stream
        for _ in range(15):
            e.process_block()
        ks = e.last_kernels()      # the 32nd block completes big block 1: X_2 behind the head
        assert ks == ([rt, "reverb_big_fft_kernel<2048,1>"] if fused else head + ["reverb_big_fft_kernel<2048,1>", rt]), ks
        e.process_block()          # TAIL(2) is there already (the side stream formed it a big block early)
        assert not any(k.startswith("reverb_big") for k in e.last_kernels())
        for _ in range(16):
            e.process_block()
        ks = e.last_kernels()      # the 49th block is the first of big block 3: TAIL(3) in front of the head
        assert ks == ["reverb_big_mac1_kernel<2048>", "reverb_big_ifft_kernel<2048,1>"] + head + [rt], ks
        e.process_block()
        assert not any(k.startswith("reverb_big") for k in e.last_kernels())
        e.close()


@pytest.mark.parametrize("B", [128, 256])
def test_small_transforms_put_off_by_calls_of_whole_big_blocks(jf, hrir, castanets, B):
    """A batch call of whole big blocks that ends on a big-block boundary puts the small transforms of its last 2 M - 1
    blocks off (state for a later call's head only); whoever takes a block through the head next forms them from the dry
    ring.  Same samples, same transform: runs of calls that mix such calls with per-block calls, ragged batch calls, calls
    that begin inside a big block, resets and a new signal must be bit-identical to an engine that forms them at once
    (jf_debug_set_reverb_lazy_state), and both within tolerance of the float64 model."""
    S = 3
    M = 16 if B <= 128 else 8
    # call sizes in blocks: whole big blocks (put off), again (the ones owed become obsolete), a block of its own (catch-up),
    # ragged calls up to a boundary, a whole big block (put off), a batch call with a ragged end (catch-up), up to a boundary
    # again, whole big blocks (put off), per-block calls (catch-up)
    sizes = [2 * M, 3 * M, 1, 1, 5, M - 7, M, 2 * M + 3, M - 3, 2 * M, 1, 1]
    K = sum(sizes)
    ir = _ir(M * B * 5 + 77, decay=3.0)
    gain = 0.6
    sigs = [castanets[5000 * s: 5000 * s + 12000 + 91 * s] for s in range(S)]
    pos = _positions(jf, S, K)
    outs = {}
    for lazy in (True, False):
        e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=max(sizes))
        e.set_reverb_lazy_state(lazy)
        for s_ in range(S):
            e.set_signal(s_, sigs[s_])
        e.set_reverb(ir, gain)
        got, b0, seen = [], 0, []
        for n, k in enumerate(sizes):
            if n == 2:
                e.reset(1)                      # between a put-off state and its catch-up ...
            if n == 10:
                e.set_signal(2, castanets[:7000])    # ... and a new signal for a source whose old samples are still owed
            if k == 1:
                e.set_latched(pos[b0])
                got.append(e.process_block()[None])
            else:
                got.append(e.process_batch(pos[b0:b0 + k]))
            seen.append(any(x == "reverb_fft_kernel<%d>@ring" % B for x in e.last_kernels()))
            b0 += k
        e.close()
        outs[lazy] = np.concatenate(got)
        # catch-ups: the calls behind a call of whole big blocks that have a block for the head (never when formed at once)
        # (a ragged batch call that begins where whole big blocks ended is one of them: call 7)
        assert seen == ([False, False, True, False, False, False, False, True, False, False, True, False] if lazy
                        else [False] * len(sizes)), seen
    assert np.abs(outs[True]).max() > 0.02
    assert np.array_equal(outs[True], outs[False])


@pytest.mark.parametrize("B,n_ir", [(128, 16 * 128 * 4 + 77), (256, 8 * 256 * 3 + 5), (128, 128 * 90 + 3), (64, 64 * 7)])
def test_next_blocks_stage_launched_ahead(jf, hrir, castanets, B, n_ir):
    """One-block calls launch the NEXT block's reverb stage behind their own spatialiser (jf_debug_set_reverb_ahead: the
    stage needs no position) and the next call launches the spatialiser alone; a new signal, a reset, a batch call, a pause,
    a new response, the callback's one-block-late ordering and switches of the stage's knobs in between take the stage back or
    leave it pending, and whatever happens the blocks must be those of an engine that runs every stage in its own call, BIT
    FOR BIT: non-uniform responses (plain heads go ahead, blocks that complete a big block or owe a TAIL do not), a long
    uniform one (the whole stage goes ahead) and a short one."""
    S = 5
    rng = np.random.default_rng(B + n_ir)
    ir = _ir(n_ir, decay=3.0)
    sigs = [castanets[3000 * s: 3000 * s + 9000 + 131 * s] for s in range(S)]
    steps = 150
    ops = rng.integers(0, 100, steps)
    outs = {}
    ahead_seen = 0
    for ahead in (True, False):
        r2 = np.random.default_rng(77)
        e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=20)
        e.set_reverb_ahead(ahead)
        for s_ in range(S):
            e.set_signal(s_, sigs[s_])
            e.set_spherical(s_, -30 + 20 * s_, 50 * s_, 0.6 + 0.2 * s_)
        e.set_reverb(ir, 0.6)
        got = []
        have_prev = False
        for t in range(steps):
            op = int(ops[t])
            if op < 6:
                e.set_signal(int(r2.integers(0, S)), castanets[int(r2.integers(0, 30000)):][: int(r2.integers(200, 6000))])
            elif op < 12:
                e.reset(int(r2.integers(0, S)))
            elif op < 18:
                k = int(r2.integers(2, 21))
                pos = jf.positions_from_spherical(np.broadcast_to(r2.integers(-40, 90, S).astype(np.float32), (k, S)),
                                                  r2.integers(0, 360, (k, S)).astype(np.float32), np.ones((k, S), np.float32))
                if have_prev:       # (a block submitted by jf_callback is still in flight: take it first)
                    rc, y = e.collect_block()
                    assert rc == 0
                    got.append(y[None])
                    have_prev = False
                got.append(e.process_batch(pos))
                continue
            elif op < 22:
                if have_prev:
                    rc, y = e.collect_block()
                    assert rc == 0
                    got.append(y[None])
                    have_prev = False
                e.set_pause(True)
                got.append(e.process_block()[None])      # silence, nothing consumed: a stage launched ahead stays pending
                e.set_pause(False)
                continue
            elif op < 25:
                e.set_reverb_async(bool(r2.integers(0, 2)))
            elif op < 27:
                if have_prev:       # (refused while a block is in flight)
                    rc, y = e.collect_block()
                    assert rc == 0
                    got.append(y[None])
                    have_prev = False
                e.set_reverb(ir, 0.6)                    # the same response again: everything starts over
            elif op < 40:
                e.set_spherical(int(r2.integers(0, S)), float(r2.integers(-40, 90)), float(r2.integers(0, 360)), 1.0)
            if op >= 90 or have_prev:                    # stretches of jf_callback (the block comes one call late)
                if have_prev and op < 90:
                    rc, y = e.collect_block()
                    assert rc == 0
                    got.append(y[None])
                    have_prev = False
                    got.append(e.process_block()[None])
                else:
                    if have_prev:
                        rc, y = e.collect_block()
                        assert rc == 0
                        got.append(y[None])
                    assert e.submit_block() == 0
                    have_prev = True
            else:
                got.append(e.process_block()[None])
        if have_prev:
            rc, y = e.collect_block()
            assert rc == 0
            got.append(y[None])
        outs[ahead] = np.concatenate(got)
        e.close()
    assert outs[True].shape == outs[False].shape and np.abs(outs[True]).max() > 0.005
    assert np.array_equal(outs[True], outs[False])


def test_nonuniform_state_changes_midstream(jf, hrir, castanets):
    """A source reset, a new signal and a pause in the middle of a run, between steps of the big partitions: the
    non-uniform engine follows the uniform one (same calls) within the float32 tolerance of two decompositions."""
    B, S = 128, 3
    ir = _ir(16 * B * 4 + 77, decay=3.0)
    P = -(-len(ir) // B)
    engines = []
    for part in (2, 1):
        e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=8)
        e.set_reverb_partitioning(part)
        for s_ in range(S):
            e.set_signal(s_, castanets[3000 * s_: 3000 * s_ + 20000])
            e.set_spherical(s_, 10 * s_, 50 * s_, 0.7)
        e.set_reverb(ir, 0.7)
        engines.append(e)
    worst = peak = 0.0
    for k in range(120):
        if k == 37:
            for e in engines:
                e.reset(1)                                  # this source starts over; the others go on
        if k == 55:
            for e in engines:
                e.set_signal(2, castanets[40000:47000])     # the old samples stay in the delay lines
        if k in (70, 75):
            for e in engines:
                e.set_pause(k == 70)                        # nothing is consumed while paused
        if k % 9 == 0:
            for e in engines:
                for s_ in range(S):
                    e.set_spherical(s_, 10 * s_, (50 * s_ + k) % 360, 0.7)
        y = [e.process_block() for e in engines]
        worst = max(worst, float(np.abs(y[0] - y[1]).max()))
        peak = max(peak, float(np.abs(y[1]).max()))
    for e in engines:
        e.close()
    assert peak > 0.02
    assert worst <= 2 * (2e-7 + 1e-7 * np.sqrt(P)) * max(1.0, peak) * S


def test_one_block_calls_with_the_big_partitions_on_the_side_stream(jf, hrir, castanets):
    """One-block calls put the big partitions' kernels on a second stream (jf_engine_reverb.cpp: run_reverb_stage): when a block
    completes a big block, its spectrum and the TAIL of the big block after the next, a whole big block early.  Against the
    same calls with everything in line (jf_debug_set_reverb_async 0: the same kernels and sums, bit for bit) and against the
    float64 model; then a run in which batch calls of ragged sizes, a reset, a new signal and a change of the switch itself
    fall between the one-block calls -- at the first, the last and a middle block of a big block."""
    B, S, K = 128, 3, 112
    ir = _ir(16 * B * 6 + 333, decay=3.0)            # a head of 32 partitions and 5 big ones behind it
    sigs = [castanets[:9000], castanets[10000:17000], castanets[20000:33000]]
    pos = _positions(jf, S, K)
    want = _model(hrir, B, S, K, ir, 0.5, sigs, pos)
    tol = (2e-7 + 1e-7 * np.sqrt(-(-len(ir) // B))) * max(1.0, np.abs(want).max()) * S
    assert np.abs(want).max() > 0.02

    def engine(on):
        e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=16)
        e.set_reverb_partitioning(2)
        e.set_reverb_async(on)
        for s_ in range(S):
            e.set_signal(s_, sigs[s_])
        e.set_reverb(ir, 0.5)
        return e

    def blocks(e, b0, n):
        out = []
        for b in range(b0, b0 + n):
            for s_ in range(S):
                e.set_spherical(s_, pos[b, s_, 0], pos[b, s_, 1], 0.5 + 0.4 * s_)
            out.append(e.process_block())
        return out

    runs = {}
    for on in (True, False):
        e = engine(on)
        runs[on] = np.array(blocks(e, 0, K))
        side = [k for k in e.last_kernels() if k.endswith("@side")]
        assert (len(side) == 3) == on, e.last_kernels()      # block 111 is the last of big block 6
        e.close()
    assert np.abs(runs[True] - want).max() <= tol
    assert np.array_equal(runs[True], runs[False])

    # one-block calls and batch calls interleaved; the switch flipped in mid-stream
    e = engine(True)
    got, b0 = [], 0
    for kind, n in (("one", 16), ("batch", 7), ("one", 9), ("batch", 16), ("one", 1), ("batch", 15), ("one", 17),
                    ("off", 0), ("one", 20), ("on", 0), ("one", 11)):
        if kind == "one":
            got += blocks(e, b0, n)
        elif kind == "batch":
            got += list(e.process_batch(pos[b0:b0 + n]))
        else:
            e.set_reverb_async(kind == "on")
        b0 += n
    e.close()
    assert b0 == K
    assert np.abs(np.array(got) - want).max() <= tol

    # a reset and a new signal while the side stream holds promises for the next big block: as the in-line engine
    outs = []
    for on in (True, False):
        e = engine(on)
        o = blocks(e, 0, 16)             # block 15 puts X_1 and TAIL(2) on the side stream ...
        o += blocks(e, 16, 3)
        e.reset(1)                       # ... of which source 1's part must not survive this
        e.set_signal(2, castanets[40000:47000])
        o += blocks(e, 19, 45)
        outs.append(np.array(o))
        e.close()
    assert np.abs(outs[0]).max() > 0.02
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("B", [64, 128])
def test_nonuniform_rings_wrap_many_times(jf, hrir, castanets, B):
    """1500 blocks -- ~94 big blocks: every ring of the non-uniform stage (the big partitions' delay line of ~25 slots, the
    dry ring, the four places of the fut ring, the small delay line, the wet ring) wraps several times -- as one-block calls
    (side stream) with ragged batch calls strewn in, against the C oracle's uniform stream form block by block."""
    S = 2
    ir = _ir(16 * B * 3 + 5, decay=3.0)
    P = -(-len(ir) // B)
    e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=40)
    o = oracle_lib.Engine(B, 512, S, hrir)
    for s_ in range(S):
        sig = castanets[5000 * s_: 5000 * s_ + 30000]
        e.set_signal(s_, sig)
        o.set_signal(s_, sig)
    e.set_reverb(ir, 0.6)
    o.set_reverb(ir, 0.6)
    assert e.reverb_partitions()[2] > 0          # non-uniform by default at this length
    rng = np.random.default_rng(77)
    tol = (2e-7 + 1e-7 * np.sqrt(P)) * S
    done = 0
    worst = peak = 0.0
    while done < 1500:
        if rng.random() < 0.1:
            k = int(rng.integers(2, 41))
            pos = np.zeros((k, S, 5), np.float32)
            for b in range(k):
                for s_ in range(S):
                    pos[b, s_] = jf.position_from_spherical(10 * s_, (done + b + 90 * s_) % 360, 0.8)
            got, want = e.process_batch(pos), o.process_batch(pos)
            done += k
        else:
            for s_ in range(S):
                e.set_spherical(s_, 10 * s_, (done + 90 * s_) % 360, 0.8)
                o.set_spherical(s_, 10 * s_, (done + 90 * s_) % 360, 0.8)
            got, want = e.process_block(), o.process_block()
            done += 1
        peak = max(peak, float(np.abs(want).max()))
        worst = max(worst, float(np.abs(got - want).max()))
        assert np.abs(got - want).max() <= tol * max(1.0, float(np.abs(want).max())), done
    e.close()
    o.close()
    assert peak > 0.05, peak
