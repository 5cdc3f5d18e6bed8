"""Engine behaviour through the C ABI on a real MI355X: the callback orderings of
Audio.cu:94-163, state handling, edge cases, and full-size runs checked through
size-independent properties."""
import os

import numpy as np
import pytest

import model64
import oracle_lib
from conftest import assert_within, sum_tol

pytestmark = pytest.mark.gpu

TOL64 = 2e-7   # the reference's own CPU-vs-GPU bound (precision_test.cu:2158)
TOL32 = 4e-7


def _workload():
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("wl", os.path.join(ROOT, "jefferson-2.0_amd", "workload.py"))
    wl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wl)
    return wl


def test_callback_has_one_block_latency(jf, hrir, castanets):
    """jf_callback = callback_func under GPU_FD_COMPLEX (Audio.cu:104-117): call k returns the
    block submitted by call k-1; the harness primes with one call (precision_test.cu:2110)."""
    a = jf.Engine(256, 512, 1, hrir=hrir)
    b = jf.Engine(256, 512, 1, hrir=hrir)
    for e in (a, b):
        e.set_signal(0, castanets)
        e.set_spherical(0, 5, 3, 0.5)
    first = a.callback()
    assert not first.any()
    for k in range(4):
        if k == 2:
            a.set_spherical(0, 5, 8, 0.5)  # latched by the submit inside THIS call -> block 3
        if k == 3:
            b.set_spherical(0, 5, 8, 0.5)  # zero-latency ordering: block 3
        want = b.process_block()
        assert np.array_equal(a.callback(), want)
    a.close()
    b.close()


def test_submit_collect_state_errors(jf, hrir):
    e = jf.Engine(128, 512, 1, hrir=hrir)
    rc, _ = e.collect_block()
    assert rc == jf.JF_ERR_STATE
    assert e.submit_block() == jf.JF_OK
    assert e.submit_block() == jf.JF_ERR_STATE
    rc, out = e.collect_block()
    assert rc == jf.JF_OK and not out.any()   # no signal set: silence
    e.close()


def test_pause_reset_and_signal_swap(jf, hrir, castanets):
    e = jf.Engine(256, 512, 2, hrir=hrir)
    o = oracle_lib.Engine(256, 512, 2, hrir)
    for x in (e, o):
        x.set_signal(0, castanets[:7000])
        x.set_signal(1, castanets[9000:9300])
        x.set_spherical(0, 20, 100, 1.0)
        x.set_spherical(1, -30, 250, 2.0)
    for _ in range(3):
        assert np.abs(e.process_block() - o.process_block()).max() <= TOL32
    e.set_pause(True)                      # Audio.cu:101: silence, nothing consumed
    assert not e.process_block().any()
    e.set_pause(False)
    assert np.abs(e.process_block() - o.process_block()).max() <= TOL32
    for x in (e, o):                       # new signal mid-stream: old samples stay in the window
        x.set_signal(1, castanets[20000:26000])
    for _ in range(3):
        assert np.abs(e.process_block() - o.process_block()).max() <= TOL32
    for x in (e, o):
        x.reset(0)
        x.reset(1)
    for _ in range(2):
        assert np.abs(e.process_block() - o.process_block()).max() <= TOL32
    e.close()


def test_setters_and_range_errors(jf, hrir):
    e = jf.Engine(256, 512, 1, hrir=hrir)
    assert e.set_spherical(0, 91, 0, 1) == jf.JF_ERR_RANGE
    assert e.set_spherical(0, -50, 0, 1) == jf.JF_ERR_RANGE
    assert e.set_spherical(1, 0, 0, 1) == jf.JF_ERR_ARG
    assert e.set_cartesian(0, 0, 0, 0) == jf.JF_ERR_RANGE
    assert e.set_cartesian(0, 0.3, 0.1, -0.4) == jf.JF_OK
    p = e.get_position(0)
    o = oracle_lib.from_cartesian(0.3, 0.1, -0.4)
    assert np.array_equal(p[:3], o) and np.array_equal(p[3:], np.float32([0.3, 0.1, -0.4]))
    assert e.set_spherical(0, 5.4, 2.6, 0.5) == jf.JF_OK      # rounds to whole degrees
    assert e.get_position(0)[:2].tolist() == [5.0, 3.0]
    e.close()


def test_cartesian_path_and_handedness(jf, hrir, castanets):
    """graphics.cu:378 drives updateFromCartesian every frame; +x maps to azimuth 270."""
    e = jf.Engine(256, 512, 1, hrir=hrir)
    o = oracle_lib.Engine(256, 512, 1, hrir)
    for x in (e, o):
        x.set_signal(0, castanets)
    for (x, y, z) in [(0.5, 0.0, 0.0), (0.45, 0.1, -0.2), (0.0, 0.3, 0.4), (-0.6, -0.2, 0.1)]:
        e.set_cartesian(0, x, y, z)
        o.set_cartesian(0, x, y, z)
        assert np.abs(e.process_block() - o.process_block()).max() <= TOL32
    e.set_cartesian(0, 1.0, 0.0, 0.0)
    assert e.get_position(0)[1] == 270.0
    e.close()


@pytest.mark.parametrize("B", [64, 128, 192, 256])
def test_block_sizes(jf, hrir, castanets, B):
    e = jf.Engine(B, 512, 1, hrir=hrir)
    m = model64.Model(B, 512, 1, hrir)
    for x in (e, m):
        x.set_signal(0, castanets[3000:])
    worst = 0.0
    for k in range(10):
        for x in (e, m):
            x.set_spherical(0, 33, (17 + 9 * k) % 360, 0.7)
        worst = max(worst, np.abs(e.process_block() - m.process_block()).max())
    e.close()
    assert worst <= TOL64


def test_distance_sweep_gain_and_delay(jf, hrir):
    """a7: gain 1/(1 + fsvs r'^2) and the fractional circular delay, for radii up to the
    alias-free limit at B = 256 (|coords| <= 5).  The reference's bound -- 2e-7 (precision_test.cu:2158), for outputs
    below 1: it flags |y| > 1 as clipping, Audio.cu:111 -- is asserted at EVERY radius on a stimulus inside that regime
    (noise of amplitude 0.35 at the nearest radius, where the path's gain is ~2; 0.5 elsewhere).  The louder stimulus at the
    nearest radius (amplitude 0.5: |y| 0.9-1.3, the clipping regime) is held to its rms error instead: float32 transforms of 1024 points
    leave 4.2e-8 rms there and the worst of a few thousand samples is 5-6 sigma, 2.0-2.8e-7 over six seeds whichever way the
    complex products are written (profiles/r03/accuracy_seeds.txt; which stage owns that floor: profiles/r04/
    error_floor.md) -- a known deviation recorded in DESIGN.md section 2: bounded below by its rms (<= 6e-8) and the worst sample's
    distance from it (<= 7 sigma)."""
    rng = np.random.default_rng(11)
    noise = rng.uniform(-.5, .5, 8192).astype(np.float32)
    for r, amp in ((0.05, 0.7), (0.05, 1.0), (0.5, 1.0), (1.0, 1.0), (2.0, 1.0), (3.5, 1.0), (4.9, 1.0)):
        sig = (noise * np.float32(amp)).astype(np.float32)
        e = jf.Engine(256, 512, 1, hrir=hrir)
        m = model64.Model(256, 512, 1, hrir)
        for x in (e, m):
            x.set_signal(0, sig)
            x.set_spherical(0, 0, 45, r)
        loud = r < 0.1 and amp >= 1.0          # noise of amplitude 0.5 at gain ~2: |y| 0.9-1.3
        sq = n = 0
        peak = worst_loud = 0.0
        for _ in range(8):
            y, y64 = e.process_block(), m.process_block()
            peak = max(peak, float(np.abs(y64).max()))
            if loud:
                worst_loud = max(worst_loud, float(np.abs(y - y64).max()))
            else:
                assert np.abs(y64).max() < 1.0
                assert np.abs(y - y64).max() <= TOL64         # the reference's bound, as it stands
            sq += float(np.sum((y - y64) ** 2))
            n += y.size
        rms = float(np.sqrt(sq / n))
        assert rms <= (6e-8 if loud else 4e-8 if r < 0.1 else 3e-8), (r, amp)     # (|y| to 0.9 at r = 0.05)
        if loud:
            # the deviation is BOUNDED, not waved through: the worst of the 4096 samples stays within 7 sigma of an error
            # whose rms is bounded above (measured 4.8-6.7 sigma over six seeds) -- i.e. below 4.2e-7 here
            assert worst_loud <= 7.0 * rms, (worst_loud, rms)
        assert peak > (0.9 if loud else 0.004), (r, amp, peak)
        e.close()


def test_wrap_short_and_empty_signals(jf, hrir):
    rng = np.random.default_rng(3)
    for n in (0, 1, 100, 255, 256, 257, 1000, 1023, 1025):
        sig = rng.uniform(-.5, .5, n).astype(np.float32)
        e = jf.Engine(256, 512, 1, hrir=hrir, max_batch_blocks=3)
        o = oracle_lib.Engine(256, 512, 1, hrir)
        pos = np.tile(jf.position_from_spherical(5, 3, 0.5), (7, 1, 1))
        for x in (e, o):
            x.set_signal(0, sig)
        y, y32 = e.process_batch(pos), o.process_batch(pos)
        e.close()
        assert np.abs(y - y32).max() <= TOL32, n
        if n == 0:
            assert not y.any()


def test_batch_fetch_completes_the_device_resident_form(jf, hrir, castanets):
    """jf_batch_upload_positions + jf_batch_run(NULL) + jf_batch_fetch (round 6: the device-resident form of callback_func's loop,
    Audio.cu:104-117, without a device pointer in the host's hands) hands out what jf_process_batch hands out, bit for bit;
    JF_ERR_STATE before any run, for more blocks than the run left, and after a run into a buffer of the caller's."""
    B, K = 256, 6
    pos = np.stack([np.stack([jf.position_from_spherical(5 * (b % 3), 10 * b + 3 * s, 0.5 + 0.1 * s) for s in range(3)])
                    for b in range(2 * K)]).astype(np.float32)
    a = jf.Engine(B, 512, 3, hrir=hrir, max_batch_blocks=K)
    b = jf.Engine(B, 512, 3, hrir=hrir, max_batch_blocks=K)
    for e in (a, b):
        for s in range(3):
            e.set_signal(s, castanets[1000 * s: 1000 * s + 30000])
    with pytest.raises(jf.JfError) as ex:
        a.batch_fetch(1)
    assert ex.value.code == jf.JF_ERR_STATE
    want = b.process_batch(pos)
    a.upload_positions(pos)
    got = []
    for k0 in (0, K):
        a.batch_run(k0, K)
        got.append(a.batch_fetch(K))
    assert np.array_equal(np.concatenate(got), want)
    a.batch_run(0, 2)
    assert a.batch_fetch(2).shape == (2, 2 * B)
    with pytest.raises(jf.JfError) as ex:
        a.batch_fetch(3)                      # the run left two blocks
    assert ex.value.code == jf.JF_ERR_STATE
    a.batch_run(2, 2, a.mix_device_ptr())     # a device pointer of the caller's (here: the engine's own, but the engine cannot know)
    with pytest.raises(jf.JfError) as ex:
        a.batch_fetch(2)
    assert ex.value.code == jf.JF_ERR_STATE
    a.close()
    b.close()


def test_invalid_batch_positions_are_silence_not_faults(jf, hrir, castanets):
    """Records a caller could hand to the batch call without going through the setters."""
    e = jf.Engine(256, 512, 3, hrir=hrir, max_batch_blocks=4)
    for s in range(3):
        e.set_signal(s, castanets[:5000])
    pos = np.tile(jf.position_from_spherical(5, 3, 0.5), (4, 3, 1))
    pos[1, 0, 0] = 95.0           # no such elevation ring
    pos[2, 1, 1] = np.nan         # NaN azimuth
    pos[3, 2, 2:] = np.inf        # infinite coordinates
    y = e.process_batch(pos)
    assert np.isfinite(y).all()
    e.close()


@pytest.mark.parametrize("B,G", [(64, 4), (128, 8), (192, 2), (256, 8), (256, 16)])
def test_group_kernel_mixed_units(jf, hrir, castanets, B, G):
    """fused_pair_kernel on units that mix everything: sources that crossfade, sources that do not (inside a
    unit that does, and in units where nothing moves), silent sources (position not interpolable), ragged
    signal lengths, the FD_BASIC mode, carried state across two calls -- against the per-source kernel
    (G = 1) and the float32 oracle."""
    S, K = 32, 6
    pos = np.zeros((2 * K, S, 5), np.float32)
    for k in range(2 * K):
        for s in range(S):
            moving = (s % 3 != 0) and not (8 <= s < 16)          # sources 8..15: a unit where nothing crossfades
            ele = -38 + (5 * s) % 120
            azi = (17 * s + (k if moving else 0) * (1 + s % 4)) % 360
            pos[k, s] = jf.position_from_spherical(ele, azi, 0.4 + 0.05 * s)
    pos[:, 5, 0] = -60.0                                           # no such elevation ring: silence
    sigs = [np.roll(castanets, 321 * s)[: 5000 + 257 * s] for s in range(S)]
    outs = {}
    for g in (1, G):
        e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=K)
        e.set_source_group(g)
        for s in range(S):
            e.set_signal(s, sigs[s])
        first = e.process_batch(pos[:K])
        e.set_mode(jf.JF_MODE_FD_BASIC)
        basic = e.process_batch(pos[K:K + 2])
        e.set_mode(jf.JF_MODE_FD_COMPLEX)
        second = e.process_batch(pos[K + 2:])
        e.close()
        outs[g] = np.concatenate([first, basic, second])
    ora = oracle_lib.Engine(B, 512, S, hrir)
    for s in range(S):
        ora.set_signal(s, sigs[s])
    want = [ora.process_batch(pos[:K])]
    ora.set_mode(1)
    want.append(ora.process_batch(pos[K:K + 2]))
    ora.set_mode(0)
    want.append(ora.process_batch(pos[K + 2:]))
    want = np.concatenate(want)
    assert np.abs(want).max() > 0.1
    assert_within(outs[G], outs[1], sum_tol(TOL32, S), f'mixed units B={B} G={G}: pair vs per-source kernel', scale=False)
    assert_within(outs[G], want, sum_tol(TOL32, S), f'mixed units B={B} G={G}: pair vs oracle32', scale=False)
    assert_within(outs[1], want, sum_tol(TOL32, S), f'mixed units B={B}: per-source kernel vs oracle32', scale=False)


def test_source_shards_sum_to_the_whole(jf, hrir, castanets):
    """sources are independent until the final sum, so a multi-GPU job is N engines over
    contiguous slices of the sources and a sum of their mixes.  Two engines with half the sources each against one
    engine with all of them (the RCCL reduce of bench.py does the addition across processes)."""
    S, K, B = 32, 12, 256
    wl = _workload()
    pos = np.zeros((K, S, 5), np.float32)
    for k in range(K):
        for s in range(S):
            pos[k, s] = jf.position_from_spherical(-35 + (9 * s) % 120, (23 * s + 2 * k) % 360, 0.5 + 0.03 * s)
    sigs = [np.roll(castanets, 911 * s)[:20000] for s in range(S)]
    whole = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=K)
    for s in range(S):
        whole.set_signal(s, sigs[s])
    want = whole.process_batch(pos)
    whole.close()
    total = np.zeros_like(want)
    for rank in range(2):
        lo, hi = wl.shard_range(S, 2, rank)
        part = jf.Engine(B, 512, hi - lo, hrir=hrir, max_batch_blocks=K)
        for s in range(lo, hi):
            part.set_signal(s - lo, sigs[s])
        total += part.process_batch(np.ascontiguousarray(pos[:, lo:hi]))
        part.close()
    assert np.abs(want).max() > 0.2
    assert_within(total, want, sum_tol(TOL32, S), 'two shards vs the whole', scale=False)


def test_full_size_moving_workload_properties(jf, hrir):
    """BASELINE.json configs[2] at full width (1024 moving sources, B = 256): the oracle is too
    slow to replay it all in a test, so check (1) a sample of sources against the oracle,
    (2) mix == ordered sum of the per-source blocks, (3) linearity in the signals."""
    wl = _workload()
    S, K, B = 1024, 8, 256
    ids = np.arange(S)
    pos = wl.trajectories(jf, ids, K)
    sigs = [wl.source_signal_and_start(s, 4096)[0] for s in ids]
    e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=K)
    e.set_source_group(1)   # per-source blocks (the reference's `intermediate`) for check (1)
    for s in ids:
        e.set_signal(int(s), sigs[s])
    e.upload_positions(pos)
    e.batch_run(0, K)
    e.synchronize()
    part = e.read_device(e.partial_device_ptr(), (K, S, 2 * B))
    mix = e.read_device(e.mix_device_ptr(), (K, 2 * B))
    e.close()

    # (1) sampled sources vs the float64 model
    sample = [0, 1, 17, 255, 511, 640, 1000, 1023]
    mod = model64.Model(B, 512, len(sample), hrir)
    for j, s in enumerate(sample):
        mod.set_signal(j, sigs[s])
    _, p64 = mod.process_batch(pos[:, sample])
    for j, s in enumerate(sample):
        assert np.abs(part[:, s] - p64[j]).max() <= TOL64, s

    # (2) the mix kernel's order: 16 groups of 64 sources, each in source order, then in group order
    acc = np.zeros((K, 2 * B), np.float32)
    for g in range(16):
        gsum = np.zeros((K, 2 * B), np.float32)
        for s in range(64 * g, 64 * (g + 1)):
            gsum = gsum + part[:, s]
        acc = gsum if g == 0 else acc + gsum
    assert np.array_equal(mix, acc)
    assert np.abs(mix - part.astype(np.float64).sum(axis=1)).max() < 2e-5  # |mix| ~ 10

    # (3) linearity: halving every signal halves the mix (exact in float: powers of two)
    e2 = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=K)
    e2.set_source_group(1)
    for s in ids:
        e2.set_signal(int(s), (0.5 * sigs[s]).astype(np.float32))
    half = e2.process_batch(pos)
    e2.close()
    assert np.array_equal(half, (0.5 * mix).astype(np.float32))

    # (4) the default grouping: G = 4 or 8 consecutive sources per wavefront, their NEW filter sets summed as
    # spectra and inverted once (fused_pair_kernel; the inverse transform is linear and the crossfade ramp
    # is common to all sources).  Same sum, other roundings: the per-group blocks against the sums of the
    # per-source blocks, and the mix against the per-source mix.
    for G in (4, 8):
        e3 = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=K)
        e3.set_source_group(G)
        for s in ids:
            e3.set_signal(int(s), sigs[s])
        grouped = e3.process_batch(pos)
        gpart = e3.read_device(e3.partial_device_ptr(), (K, S // G, 2 * B))
        e3.close()
        want = part.astype(np.float64).reshape(K, S // G, G, 2 * B).sum(axis=2)
        assert_within(gpart, want, sum_tol(TOL64, G), f'full width: groups of {G} vs sums of per-source blocks', scale=False)
        assert np.abs(grouped - mix).max() < 2e-5, G           # |mix| ~ 10


def _write_compact_dir(root, hrir):
    """Re-create the reference's compact KEMAR layout (compact/elev%d/H%de%03da.wav, stereo int16)
    from the committed table fixture, so the directory loader can be exercised off the build box."""
    import wave
    for j, (e, a) in enumerate(model64.table_positions()):
        if a > 180:
            continue
        d = os.path.join(root, f"elev{e}")
        os.makedirs(d, exist_ok=True)
        pcm = np.round(hrir[j].T * 32768.0).astype(np.int16)  # [128][2], ch0 = left
        with wave.open(os.path.join(d, f"H{e}e{a:03d}a.wav"), "wb") as w:
            w.setnchannels(2)
            w.setsampwidth(2)
            w.setframerate(44100)
            w.writeframes(pcm.tobytes())


def test_directory_loader_and_offline_driver(jf, hrir, castanets, tmp_path):
    """jf_engine_create_from_dir on a compact-layout directory, and the plain-C offline driver
    (jf_render: WAV in -> stereo 24-bit WAV out, benchmarkTesting trajectory) against the oracle."""
    import struct
    import subprocess
    import wave
    from conftest import ROOT, scenario_positions
    kemar = str(tmp_path / "compact")
    _write_compact_dir(kemar, hrir)
    eng = jf.Engine(256, 512, 1, hrir_dir=kemar)
    ref_table = jf.Engine(256, 512, 1, hrir=hrir).read_table()
    assert np.array_equal(eng.read_table(), ref_table)
    eng.close()

    # the reference's own "full" layout (full/elev%d/L%de%03da.wav + R..., mono files, hrtf_signals.cu:124,131)
    full = str(tmp_path / "full")
    for j, (e, a) in enumerate(model64.table_positions()):
        d = os.path.join(full, f"elev{e}")
        os.makedirs(d, exist_ok=True)
        for ear, tag in enumerate("LR"):
            with wave.open(os.path.join(d, f"{tag}{e}e{a:03d}a.wav"), "wb") as w:
                w.setnchannels(1)
                w.setsampwidth(2)
                w.setframerate(44100)
                w.writeframes(np.round(hrir[j, ear] * 32768.0).astype(np.int16).tobytes())
    eng = jf.Engine(256, 512, 1, hrir_dir=full)
    assert np.array_equal(eng.read_table(), ref_table)
    eng.close()
    # a stereo file where the full layout wants mono is the reference's "incorrect number of channels"
    bad = os.path.join(full, "elev0", "L0e000a.wav")
    with wave.open(bad, "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(44100)
        w.writeframes(np.zeros((128, 2), np.int16).tobytes())
    with pytest.raises(jf.JfError) as ei:
        jf.Engine(256, 512, 1, hrir_dir=full)
    assert ei.value.code == jf.JF_ERR_IO

    ex = np.load(os.path.join(ROOT, "tests", "golden", "castanets_441_excerpt_i24.npy"))[:30000]
    inp = str(tmp_path / "in.wav")
    with wave.open(inp, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(3)
        w.setframerate(44100)
        w.writeframes(b"".join(struct.pack("<i", int(v))[:3] for v in ex))
    outp = str(tmp_path / "out.wav")
    exe = os.path.join(ROOT, "jefferson-2.0_amd", "jf_render")
    r = subprocess.run([exe, kemar, inp, outp, "--block", "256", "--azi", "3", "--ele", "5",
                        "--dwell", "4", "--rounds", "5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    with wave.open(outp) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getnframes()) == (2, 3, 256 * 4 * 6)
        raw = np.frombuffer(w.readframes(w.getnframes()), np.uint8).reshape(-1, 3).astype(np.int32)
    v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
    got = (np.where(v >= 1 << 23, v - (1 << 24), v) / 8388607.0).reshape(-1, 512)

    ora = oracle_lib.Engine(256, 512, 1, hrir)
    ora.set_signal(0, castanets[:30000])
    ora.reset(0)
    want = []
    for (ele, azi, rr) in scenario_positions(3, 5, 4, 5):
        ora.set_spherical(0, ele, azi, rr)
        want.append(ora.process_block())
    assert np.abs(got - np.array(want)).max() <= 1.0 / 8388607 + TOL32

    # --batch: the same trajectory handed over as latched records, 7 blocks per call (ragged last call)
    outb = str(tmp_path / "outb.wav")
    r = subprocess.run([exe, kemar, inp, outb, "--block", "256", "--azi", "3", "--ele", "5",
                        "--dwell", "4", "--rounds", "5", "--batch", "7"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(outb, "rb").read() == open(outp, "rb").read()


def test_scripted_motion_run_debugmode2(jf, hrir, castanets, tmp_path):
    """`jf_render --script debugmode2` = the audio-only build's scripted run (DEBUGMODE 2, main.cu:101-149): the source
    starts at the constructor's (0, 0) and takes the way-points (ele, azi) = (4,2), (3,1), (2,4), (9,7), (0,0), the k-th as soon
    as its play position has reached (k * 44100) % length, then two more seconds.  Per-block calls (the setter before every
    block, as the main thread does) and --batch (the same latched records up front) must write the same file, and that file
    is the C oracle's output for the same script at the float32 tolerance (+ one 24-bit step); the five positions
    interpolate as tests/golden/interp_known.json says."""
    import json
    import struct
    import subprocess
    import wave
    from conftest import ROOT
    kemar = str(tmp_path / "compact")
    _write_compact_dir(kemar, hrir)
    B = 256
    # 6.2 s of input: the 2 s excerpt three times over, each pass with another gain and a slow fade, + a bit more
    x = np.concatenate([castanets * g for g in (0.9, 0.6, 0.8)] + [castanets[:9000] * 0.5]).astype(np.float32)
    x *= (0.75 + 0.25 * np.cos(np.arange(len(x)) / 30000.0)).astype(np.float32)
    pcm = np.round(x.astype(np.float64) * 8388607.0).astype(np.int64)
    inp = str(tmp_path / "in.wav")
    with wave.open(inp, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(3)
        w.setframerate(44100)
        w.writeframes(b"".join(struct.pack("<i", int(v))[:3] for v in pcm))
    sig, fs = jf.wav_read_mono(inp)             # what the driver itself reads (libsndfile scaling: / 2^23)
    assert fs == 44100 and len(sig) == len(x)
    exe = os.path.join(ROOT, "jefferson-2.0_amd", "jf_render")

    def render(extra):
        outp = str(tmp_path / ("out_" + "_".join(extra).replace("-", "") + ".wav"))
        r = subprocess.run([exe, kemar, inp, outp, "--block", str(B), "--script", "debugmode2"] + extra, capture_output=True,
                           text=True)
        assert r.returncode == 0 and "debugmode2" in r.stderr, r.stderr
        with wave.open(outp) as w:
            assert (w.getnchannels(), w.getsampwidth()) == (2, 3)
            raw = np.frombuffer(w.readframes(w.getnframes()), np.uint8).reshape(-1, 3).astype(np.int32)
        v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
        return (np.where(v >= 1 << 23, v - (1 << 24), v) / 8388607.0).reshape(-1, 2 * B), open(outp, "rb").read()

    got, raw_block = render([])
    got_b, raw_batch = render(["--batch", "96"])
    assert raw_batch == raw_block
    # the script, restated here: way-point k latched by the first block that starts with count >= (k * 44100) % length
    way = [(4, 2), (3, 1), (2, 4), (9, 7), (0, 0)]
    n, count, k, pos, left = len(sig), 0, 1, (0, 0), None
    script = []
    while True:
        while k <= 5 and count >= (k * 44100) % n:
            pos = way[k - 1]
            k += 1
            if k == 6:
                left = -(-2 * 44100 // B)
        if k == 6:
            if left == 0:
                break
            left -= 1
        script.append(pos)
        count = count + B if count + B < n else B - (n - count)
    assert len(script) == got.shape[0]
    first = [script.index(p) for p in way[:4]]
    first.append(first[3] + script[first[3]:].index((0, 0)))
    assert first == [-(-44100 * (j + 1) // B) for j in range(5)]      # one way-point per second of consumed input
    ora = oracle_lib.Engine(B, 512, 1, hrir)
    ora.set_signal(0, sig)
    ora.reset(0)
    want = []
    for (ele, azi) in script:
        ora.set_spherical(0, ele, azi, 0.5)
        want.append(ora.process_block())
    want = np.array(want)
    assert np.abs(want).max() > 0.05
    assert np.abs(got - want).max() <= 1.0 / 8388607 + TOL32
    # the blocks right after a way-point differ from a run that stays at (0, 0): the script really moves the source
    assert np.abs(got[first[0]] - got[first[0] - 1]).max() > 0
    known = json.load(open(os.path.join(ROOT, "tests", "golden", "interp_known.json")))["points"]
    for (ele, azi) in way:
        idx, om = jf.interpolation(float(ele), float(azi))
        assert idx.tolist() == known[f"{ele},{azi}"]["idx"]
        assert np.array_equal(om, np.float32(known[f"{ele},{azi}"]["omegas"]))


def test_rccl_reduce_on_the_engine_stream():
    """The RCCL side of bench.py's N > 1 path on the one GPU of this box: the `nccl` backend with world_size 1
    (communicator init, the engine's stream wrapped as an ExternalStream, asynchronous dist.reduce ordered after
    the kernels, double-buffered reuse).  A reduce over one rank must hand back exactly the engine's mix."""
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["JF_ROOT"])
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from jf_load import jf
import importlib.util
spec = importlib.util.spec_from_file_location("wl", os.path.join(os.environ["JF_ROOT"], "jefferson-2.0_amd", "workload.py"))
wl = importlib.util.module_from_spec(spec); spec.loader.exec_module(wl)
hrir = np.load(os.path.join(os.environ["JF_ROOT"], "tests", "golden", "kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
S, KB, B, STEPS = 64, 8, 256, 6
ids = np.arange(S)
def engine():
    e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=KB)
    for s in ids:
        e.set_signal(int(s), wl.source_signal_and_start(s, 9000)[0])
    e.upload_positions(wl.trajectories(jf, ids, KB * STEPS))
    return e
ref = engine()
want = []
for i in range(STEPS):
    ref.batch_run(i * KB, KB)
    ref.synchronize()
    want.append(ref.read_device(ref.mix_device_ptr(), (KB, 2 * B)))
ref.close()
eng = engine()
ext = torch.cuda.ExternalStream(eng.stream_ptr())
mixes = [torch.zeros((KB, 2 * B), dtype=torch.float32, device="cuda") for _ in range(2)]
pending = [None, None]
got = []
for i in range(STEPS):
    j = i & 1
    if pending[j] is not None:
        with torch.cuda.stream(ext):
            pending[j][0].wait()
        pending[j][0].wait()
        torch.cuda.synchronize()
        got.append((pending[j][1], mixes[j].cpu().numpy().copy()))
        pending[j] = None
    eng.batch_run(i * KB, KB, mixes[j].data_ptr())
    with torch.cuda.stream(ext):
        pending[j] = (dist.reduce(mixes[j], dst=0, op=dist.ReduceOp.SUM, async_op=True), i)
for j in range(2):
    if pending[j] is not None:
        pending[j][0].wait()
        torch.cuda.synchronize()
        got.append((pending[j][1], mixes[j].cpu().numpy().copy()))
eng.synchronize()
dist.barrier()
torch.cuda.synchronize()
assert sorted(i for i, _ in got) == list(range(STEPS))
for i, m in got:
    assert np.abs(want[i]).max() > 0.1
    assert np.array_equal(m, want[i]), i
eng.close()
dist.destroy_process_group()
print("RCCL_OK", dist.is_nccl_available())
'''
    env = dict(os.environ, JF_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    assert "RCCL_OK True" in r.stdout.decode()


def test_group_of_one_gpu_equals_the_engine(jf, hrir, castanets):
    """include/jefferson_group.h on the one GPU of this box: a communicator of size 1 (ncclCommInitAll, ncclReduce on
    the engine's stream behind its kernels).  Batch calls in several runs, the device-resident form and per-block
    calls must hand back exactly what a single engine produces."""
    import importlib
    grp = importlib.import_module("jefferson_amd.group")
    S, K, B = 24, 20, 256
    pos = np.zeros((K, S, 5), np.float32)
    for k in range(K):
        for s in range(S):
            pos[k, s] = jf.position_from_spherical(-35 + (9 * s) % 120, (23 * s + 2 * k) % 360, 0.5 + 0.03 * s)
    sigs = [np.roll(castanets, 911 * s)[:20000 + 10 * s] for s in range(S)]
    eng = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=8)
    g = grp.Group(B, 512, S, hrir, n_gpus=1, max_batch_blocks=8)
    assert g.num_gpus() == 1 and g.first_source(0) == 0
    for s in range(S):
        eng.set_signal(s, sigs[s])
        g.set_signal(s, sigs[s])
    want = eng.process_batch(pos)          # 20 blocks = runs of 8, 8, 4
    got = g.process_batch(pos)
    assert np.abs(want).max() > 0.2
    assert np.array_equal(got, want)
    # device-resident form, state carried on from the batch above
    more = np.ascontiguousarray(pos[::-1])
    eng.upload_positions(more)
    g.upload_positions(more)
    for first, n in ((0, 8), (8, 5)):
        eng.batch_run(first, n)
        eng.synchronize()
        w = eng.read_device(eng.mix_device_ptr(), (n, 2 * B))
        assert g.batch_run(first, n) == jf.JF_OK
        assert g.batch_run(first, n) == jf.JF_ERR_STATE      # one run in flight
        rc, out = g.batch_fetch(n)
        assert rc == jf.JF_OK and np.array_equal(out, w)
    assert g.batch_fetch(1)[0] == jf.JF_ERR_STATE
    # per-block calls (host sum over the shards)
    for k in range(3):
        for s in range(S):
            assert g.set_spherical(s, 10, (30 * s + 5 * k) % 360, 1.0) == jf.JF_OK
            assert eng.set_spherical(s, 10, (30 * s + 5 * k) % 360, 1.0) == jf.JF_OK
        assert np.array_equal(g.process_block(), eng.process_block())
    assert g.set_spherical(S, 0, 0, 1.0) == jf.JF_ERR_ARG
    # the job-wide controls: reverb stage, nearest-HRTF mode, pause, reset -- forwarded to every engine
    ir = (np.random.default_rng(8).standard_normal(700) * np.exp(-np.arange(700) / 150.0) * 0.1).astype(np.float32)
    eng.set_reverb(ir, 0.9)
    g.set_reverb(ir, 0.9)
    for k in range(10):
        if k in (3, 6):
            eng.set_mode(jf.JF_MODE_FD_BASIC if k == 3 else jf.JF_MODE_FD_COMPLEX)
            assert g.set_mode(jf.JF_MODE_FD_BASIC if k == 3 else jf.JF_MODE_FD_COMPLEX) == jf.JF_OK
        if k in (7, 8):
            eng.set_pause(k == 7)
            g.set_pause(k == 7)
        if k == 9:
            eng.reset(5)
            g.reset(5)
        a, b = g.process_block(), eng.process_block()
        assert np.array_equal(a, b)
        assert g.last_block_peak() == eng.last_block_peak() == float(np.abs(b).max())
        assert (np.abs(b).max() == 0) == (k == 7)
    assert g.set_mode(5) == jf.JF_ERR_ARG and g.failed() == 0
    g.close()
    eng.close()


@pytest.mark.parametrize("args", [["pa"], ["group", "1"], ["shards", "2"], ["shards", "3"], ["bench", "1", "24"]])
def test_plain_c_boundary_checks(args):
    """jf_ctest.c (plain C, linked against the C ABI only): `pa` drives jf_pa_callback with PortAudio's argument list
    for 200 blocks -- positions and pause changed in between -- against jf_callback on a twin engine (paCallback,
    Audio.cu:164-175); `group 1` runs jefferson_group.h over one GPU (RCCL communicator of size 1) against one engine,
    then again with the job-wide controls (reverb stage, mode switch, pause, reset, clip peak: Audio.cu:101,104,111-113,
    cudaPart.cu:65-205); `shards N` runs N shards of one job on the one device (jf_group_create_shards_on_device: the
    production sharding, the per-shard repack of the trajectory, the controls and the FAILED transitions under forced failures
    on the last shard, with a host sum where several GPUs have ncclReduce; the mixing loop it distributes: Audio.cu:109-110);
    `bench 1` drives the bench workload (1024 moving sources, 128 blocks of 256 per run) through
    jf_group_batch_run / _fetch and prints bench.py's metric."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "jefferson-2.0_amd", "jf_ctest")
    r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, (r.stdout.decode(), r.stderr.decode())
    if args[0] == "bench":
        import re
        m = re.search(rb"([0-9.e+]+) source-frames/s", r.stdout)
        assert m and float(m.group(1)) > 2e10, r.stdout      # a C host gets the kernels' throughput, not a toy figure
    else:
        assert b"max" in r.stdout
    if args[0] == "group":
        assert b"group controls" in r.stdout
    if args[0] == "shards":
        assert b"forced failures on shard" in r.stdout and b"as specified" in r.stdout


def test_pair_hand_off_time_out_is_reported_not_hung():
    """The batch kernel's waits between the two wavefronts of a pair are bounded.  With the fault-injection build of
    the library (`make -C jefferson-2.0_amd/csrc faultlib`: tests/build/libjefferson_hip_droppub.so = -DJF_EXP_DROP_PUBLISH,
    not part of the product build and refused by jf_engine_create unless JF_ALLOW_EXPERIMENT=1; jf_experiments.h: one wavefront of
    the grid stops announcing its hand-offs) the partner's wait must run out, the kernel must drain, and the engine
    must say so: JF_ERR_DEVICE with the hand-off text from jf_synchronize, from a per-block call that takes the pair
    kernel, and from everything afterwards; destroying the engine works.  One run, in a child process of its own
    (the library is chosen when the binding is loaded)."""
    import subprocess
    import sys
    from conftest import ROOT
    lib = os.path.join(ROOT, "tests", "build", "libjefferson_hip_droppub.so")
    assert os.path.exists(lib), "make -C jefferson-2.0_amd/csrc faultlib builds it (__graft_entry__.build() does)"
    # without the permission the library must refuse to make an engine at all
    refuse = ('import os, sys\nsys.path.insert(0, os.environ["JF_ROOT"])\nfrom jf_load import jf\nimport numpy as np\n'
              'try:\n    jf.Engine(256, 512, 1, hrir=np.zeros((710, 2, 128), np.float32))\n    print("CREATED")\n'
              'except jf.JfError as ex:\n    print("REFUSED", ex.code, ex)\n')
    env0 = {k: v for k, v in os.environ.items() if k != "JF_ALLOW_EXPERIMENT"}
    r0 = subprocess.run([sys.executable, "-c", refuse], env=dict(env0, JF_ROOT=ROOT, JF_LIB=lib), stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=300)
    assert b"undefined symbol" not in r0.stderr, "stale fault-injection build: make -C jefferson-2.0_amd/csrc faultlib"
    assert r0.returncode == 0 and b"REFUSED -5" in r0.stdout and b"JF_ALLOW_EXPERIMENT" in r0.stdout, (r0.stdout, r0.stderr[-1000:])
    code = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ["JF_ROOT"])
from jf_load import jf
hrir = np.load(os.path.join(os.environ["JF_ROOT"], "tests", "golden", "kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
rng = np.random.default_rng(1)
S, K, B = 64, 8, 256
pos = np.zeros((K, S, 5), np.float32)
for k in range(K):
    for s in range(S):
        pos[k, s] = jf.position_from_spherical(0, (5 * s + k) % 360, 1.0)
# --- batch call through the pair kernel
e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=K)
for s in range(S):
    e.set_signal(s, rng.uniform(-.5, .5, 5000).astype(np.float32))
e.set_source_group(8)
e.upload_positions(pos)
t0 = time.perf_counter()
e.batch_run(0, K)
try:
    e.synchronize()
    print("NO_ERROR_FROM_SYNCHRONIZE")
except jf.JfError as ex:
    print("SYNC", ex.code, str(ex))
print("SECONDS %.3f" % (time.perf_counter() - t0))
assert any("fused_pair_kernel" in k for k in e.last_kernels())
for what, call in (("RUN", lambda: e.batch_run(0, K)), ("BLOCK", e.process_block), ("BATCH", lambda: e.process_batch(pos))):
    try:
        call()
        print(what, "NO_ERROR")
    except jf.JfError as ex:
        print(what, ex.code)
e.close()
# --- per-block calls that take the pair kernel (more sources than the one-launch kernel is allowed)
e = jf.Engine(B, 512, S, hrir=hrir)
for s in range(S):
    e.set_signal(s, rng.uniform(-.5, .5, 5000).astype(np.float32))
e.set_rt_max_sources(0)
e.set_source_group(8)
try:
    e.process_block()
    print("COLLECT NO_ERROR")
except jf.JfError as ex:
    print("COLLECT", ex.code, str(ex))
assert any("fused_pair_kernel" in k for k in e.last_kernels())
out = np.ones(2 * B, np.float32)
rc = jf.lib().jf_pa_callback(None, out.ctypes.data_as(jf.C.c_void_p), B, None, 0, e.h)
print("PA", rc, float(np.abs(out).max()))
e.close()
print("CLOSED")
'''
    env = dict(os.environ, JF_ROOT=ROOT, JF_LIB=lib, JF_ALLOW_EXPERIMENT="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0, (out, r.stderr.decode(errors="replace")[-2000:])
    lines = dict(l.split(" ", 1) for l in out.splitlines() if " " in l)
    assert lines["SYNC"].startswith("-3 ") and "hand-off timed out" in lines["SYNC"], out
    assert float(lines["SECONDS"]) < 2.0, out        # one bounded wait (~0.1 s), not a hang
    assert lines["RUN"] == "-3" and lines["BLOCK"] == "-3" and lines["BATCH"] == "-3", out   # fatal from then on
    assert lines["COLLECT"].startswith("-3 ") and "hand-off timed out" in lines["COLLECT"], out
    assert lines["PA"] == "0 0.0", out               # PortAudio gets silence, never garbage
    assert "CLOSED" in out


def test_source_order_contract_when_a_run_resolves_to_single_sources(jf, hrir, castanets):
    """Automatic grouping sorts the sources by table row (jf_debug_source_order) for the pair kernel.  A run that
    resolves to G = 1 -- an odd number of sources, or a small call -- goes through the per-source kernel, whose block u
    IS source u: the reported order is then the identity, and the blocks match the oracle source by source."""
    B = 256
    for S, K in ((7, 6), (12, 2)):
        pos = np.zeros((K, S, 5), np.float32)
        for k in range(K):
            for s in range(S):
                pos[k, s] = jf.position_from_spherical(60 - 9 * s, (301 * s + 2 * k) % 360, 0.6 + 0.1 * s)   # rows far apart
        e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=K)
        ora = oracle_lib.Engine(B, 512, S, hrir)
        for s in range(S):
            sig = np.random.default_rng(s).uniform(-0.5, 0.5, 15000).astype(np.float32)
            e.set_signal(s, sig)
            ora.set_signal(s, sig)
        e.upload_positions(pos)
        e.batch_run(0, K)
        e.synchronize()
        assert e.last_source_group() == 1
        assert np.array_equal(e.source_order(), np.arange(S))
        part = e.read_device(e.partial_device_ptr(), (K, S, 2 * B))
        e.close()
        _, opart = ora.process_batch(pos, want_partial=True)
        ora.close()
        for s in range(S):
            assert np.abs(opart[s]).max() > 1e-3
            assert np.abs(part[:, s] - opart[s]).max() <= 4e-7 * max(1.0, np.abs(opart[s]).max()), (S, s)


def test_whole_bench_under_the_launcher_with_one_rank():
    """bench.py exactly as the driver starts it for N > 1 -- `python -m torch.distributed.run --nproc-per-node=N
    bench.py --gpus N` -- with N = 1, the only N this box has: the ranks are started before anything touches the GPU,
    the process group is RCCL, the mix goes through the asynchronous reduce, and rank 0 prints the line with
    `verified` and a CPU baseline on every host core this process may use (the launcher exports OMP_NUM_THREADS=1)."""
    import json
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OMP_NUM_THREADS")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", "--no-pmc"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["verified"] is True, out.get("verification")
    assert out["n_gpus"] == 1 and out["steps"] == 8
    assert out["comm"]["backend"] == "RCCL"
    # the communication fraction of SURVEY.md 8(d) config 4 is MEASURED (events around the collective), also with one rank
    assert out["comm"]["ms_per_collective"] > 0 and 0 < out["comm"]["fraction"] < 1.0, out["comm"]
    assert out["roofline"]["launches_timed"] == 8      # a short run is timed at a shorter stride: never fewer than eight launches
    assert "prewarm_policy" in out and "frac_reference_algorithm" not in out["roofline"]
    n_cpu = len(os.sched_getaffinity(0))
    assert out["cpu_baseline"]["cores"] > 1 or n_cpu == 1
    assert out["roofline"]["frac"] > 0.05 and out["value"] > 1e10
    # the secondary configurations of the default line (round 6): config 5 in batch form with its HBM roofline, verified against
    # the oracle's reverb stage; config 5 one block per call; the stationary variant -- each on a fresh engine behind the headline
    also = out["also"]
    rv = also["reverb"]
    assert rv["verified"] is True and rv["steps"] >= 32 and rv["config"]["blocks_per_step"] == 256, rv
    assert rv["roofline"]["bound"] == "hbm" and 0.2 < rv["roofline"]["frac"] < 1.0 and rv["roofline"]["avg_stage_ms"] > 0
    assert rv["roofline"]["algorithmic_bytes_per_step"] > 4e8 and "reverb_big_mac_kernel" in rv["roofline"]["kernel"]
    assert rv["value"] > 5e9 and rv["cpu_baseline"]["value"] > 0
    assert also["reverb_realtime_us"]["calls"] == 1600 and 5 < also["reverb_realtime_us"]["median"] < 200
    st = also["stationary"]
    assert st["verified"] is True and st["value"] > out["value"] and st["config"]["interp_table"]["rows_read_by_the_timed_runs"] == 1


def test_a_live_engine_survives_bad_arguments_and_keeps_working(hrir):
    """With a real engine behind the handle: every entry point called with null buffers, zero and negative counts and
    out-of-range indices must come back with an error code (or do nothing) -- and the engine must render the next block as
    if nothing had happened (against the oracle).  In a child process, each name printed before its call."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import ctypes as C, sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
from jf_load import jf
import oracle_lib
hrir = np.load(os.path.join(%r, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
e = jf.Engine(256, 512, 3, hrir=hrir, max_batch_blocks=4)
o = oracle_lib.Engine(256, 512, 3, hrir)
rng = np.random.default_rng(1)
for s in range(3):
    sig = rng.uniform(-0.5, 0.5, 5000).astype(np.float32)
    e.set_signal(s, sig); o.set_signal(s, sig)
    e.set_spherical(s, 10 * s, 40 * s, 1.0); o.set_spherical(s, 10 * s, 40 * s, 1.0)
a, b = e.process_block(), o.process_block()
assert np.abs(a - b).max() < 2e-6
L = jf.lib()
skip = {"jf_engine_destroy", "jf_engine_create", "jf_engine_create_from_dir", "jf_free"}
n = 0
for variant in (0, 1, 2):
    for name, (res, args) in sorted(jf._SIGS.items()):
        if name in skip or not args or args[0] is not C.c_void_p:
            continue
        vals = [e.h]
        for t in args[1:]:
            if t in (C.c_int, C.c_uint, C.c_long, C.c_longlong, C.c_size_t, C.c_ulong):
                vals.append(t([0, -1 if t in (C.c_int, C.c_long, C.c_longlong) else 0, 1 << 20][variant]))
            elif t in (C.c_float, C.c_double):
                vals.append(t([0.0, float("nan"), 1e30][variant]))
            else:
                vals.append(None)
        print("calling", name, variant, flush=True)
        getattr(L, name)(*vals)
        n += 1
print("CALLS", n, flush=True)
# whatever state the setters-with-garbage left (a refused call leaves none; an accepted one -- set_mode(0), set_pause(0),
# debug knobs at 0 -- is undone here), the engine still renders
tmp = np.zeros(512, np.float32)
L.jf_collect_block(e.h, tmp.ctypes.data_as(jf._f))      # jf_submit_block(e) is a valid call: a block may be in flight
L.jf_set_pause(e.h, 0); L.jf_set_mode(e.h, 0); L.jf_profile_enable(e.h, 0)
e.set_rt_max_sources(8192); e.set_interp_table(2); e.set_prep_ahead(True); e.set_source_group(0) if hasattr(e, "set_source_group") else None
L.jf_reverb_set_ir(e.h, None, 0, C.c_float(1.0))
for s in range(3):
    e.reset(s); o.reset(s)
    sig = rng.uniform(-0.5, 0.5, 5000).astype(np.float32)
    e.set_signal(s, sig); o.set_signal(s, sig)
    e.set_spherical(s, 10 * s, 40 * s, 1.0); o.set_spherical(s, 10 * s, 40 * s, 1.0)
for k in range(3):
    a, b = e.process_block(), o.process_block()
    assert np.abs(a - b).max() < 2e-6, (k, np.abs(a - b).max())
e.close()
print("STILL WORKS")
''' % (ROOT, ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "STILL WORKS" in out, (out[-800:], r.stderr.decode()[-1500:])


def test_a_live_group_survives_bad_arguments_and_keeps_working(hrir):
    """The same for include/jefferson_group.h with a live group (three shards on the one device) behind the handle."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import ctypes as C, sys, os, importlib
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
from jf_load import jf
import oracle_lib
grp = importlib.import_module("jefferson_amd.group")
hrir = np.load(os.path.join(%r, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
S = 7
g = grp.Group(256, 512, S, hrir, max_batch_blocks=4, shards_on_device=3)
o = oracle_lib.Engine(256, 512, S, hrir)
rng = np.random.default_rng(2)
def fresh():
    for s in range(S):
        g.reset(s); o.reset(s)
        sig = rng.uniform(-0.5, 0.5, 5000).astype(np.float32)
        g.set_signal(s, sig); o.set_signal(s, sig)
        g.set_spherical(s, 10 * s - 30, 40 * s, 1.0); o.set_spherical(s, 10 * s - 30, 40 * s, 1.0)
fresh()
assert np.abs(g.process_block() - o.process_block()).max() < 4e-6
L = grp.lib()
skip = {"jf_group_destroy", "jf_group_create", "jf_group_create_grid", "jf_group_create_shards_on_device", "jf_group_debug_fail_next"}
n = 0
for variant in (0, 1, 2):
    for name, (res, args) in sorted(grp._SIGS.items()):
        if name in skip or not args or args[0] is not C.c_void_p:
            continue
        vals = [g.h]
        for t in args[1:]:
            if t in (C.c_int, C.c_uint, C.c_long, C.c_longlong, C.c_size_t, C.c_ulong):
                vals.append(t([0, -1 if t in (C.c_int, C.c_long, C.c_longlong) else 0, 1 << 20][variant]))
            elif t in (C.c_float, C.c_double):
                vals.append(t([0.0, float("nan"), 1e30][variant]))
            else:
                vals.append(None)
        print("calling", name, variant, flush=True)
        getattr(L, name)(*vals)
        n += 1
print("CALLS", n, flush=True)
assert not g.failed()
L.jf_group_set_pause(g.h, 0); L.jf_group_set_mode(g.h, 0)
g.set_reverb(np.zeros(0, np.float32))
fresh()
for k in range(3):
    a, b = g.process_block(), o.process_block()
    assert np.abs(a - b).max() < 4e-6, (k, np.abs(a - b).max())
g.close()
print("STILL WORKS")
''' % (ROOT, ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "STILL WORKS" in out, (out[-800:], r.stderr.decode()[-1500:])


def test_engines_come_and_go_without_leaking_device_memory(jf, hrir, castanets):
    """Twenty engines created, used (batch and per-block calls, the reverb with its side stream, a new response) and
    destroyed: the device's free memory afterwards is what it was before the first (each engine holds ~0.4 GB of tables and
    rings; a leak of any of its buffers, streams or events would show)."""
    import torch
    torch.cuda.init()
    ir = (np.random.default_rng(3).standard_normal(16 * 128 * 3 + 5) * 0.02).astype(np.float32)

    def cycle():
        e = jf.Engine(128, 512, 24, hrir=hrir, max_batch_blocks=8)
        for s in range(24):
            e.set_signal(s, castanets[100 * s:100 * s + 4000])
        e.set_reverb(ir, 0.5)
        for _ in range(18):
            e.process_block()
        pos = np.zeros((8, 24, 5), np.float32)
        for k in range(8):
            for s in range(24):
                pos[k, s] = jf.position_from_spherical(0, (10 * s + k) % 360, 1.0)
        e.process_batch(pos)
        e.set_reverb(ir[:700], 0.5)
        e.process_block()
        e.close()
    cycle()                                  # (the first one also pays for the runtime's own pools)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(20):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, (free0, free1)
