"""CPU tests of the C-ABI library: it loads, exports every symbol include/jefferson.h
declares, its host-side logic (geometry, index rules, WAV/HRIR readers) agrees with the
oracle, and it fails loudly -- never falls back -- when no GPU is present."""
import ctypes
import os
import re
import struct
import wave

import numpy as np
import pytest

import model64
import oracle_lib
from conftest import GOLD, ROOT


def _strip_comments(src):
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def _header_functions(name="jefferson.h"):
    src = _strip_comments(open(os.path.join(ROOT, "include", name)).read())
    return sorted(set(re.findall(r"\b(jf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(jf):
    public, debug = _header_functions(), _header_functions("jefferson_debug.h")
    debug = [n for n in debug if not n.startswith("jf_group_")]   # libjefferson_group.so's test support: its own test below
    assert len(public) >= 35 and not set(public) & set(debug)
    L = ctypes.CDLL(jf.LIB_PATH)
    missing = [n for n in public + debug if not hasattr(L, n)]
    assert not missing, missing
    # and the binding covers both headers
    assert sorted(jf.exported_symbols()) == sorted(public + debug)


def test_public_header_is_the_drop_in_boundary_only():
    """include/jefferson.h holds what INTEGRATION.md binds and nothing else: no jf_debug_* / jf_profile_*
    export, no device pointer or stream accessor, nothing read from the environment by the product."""
    names = _header_functions()
    assert not [n for n in names if n.startswith(("jf_debug_", "jf_profile_"))]
    assert not {"jf_engine_stream", "jf_batch_mix_device", "jf_batch_partial_device"} & set(names)
    dbg = _header_functions("jefferson_debug.h")
    assert "jf_engine_stream" in dbg and "jf_profile_enable" in dbg and "jf_debug_stage_taps" in dbg
    csrc = os.path.join(ROOT, "jefferson-2.0_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".cpp", ".c", ".hip", ".h")) and f not in ("jf_ctest.c", "jf_render.c"):
            for m in re.findall(r'getenv\("(\w+)"\)', open(os.path.join(csrc, f)).read()):
                assert m == "JF_ALLOW_EXPERIMENT", (f, m)   # the refusal of wrong-result builds, not a tuning knob


_CITE = re.compile(r"\b[A-Za-z_]+\.(?:cuh?|cpp|md|vcxproj|sh):\d+")


def test_header_cites_reference_interfaces():
    """EVERY function of the public header stands under a comment that names the reference interface it replaces (file:line,
    relative to Jefferson/src/): the comment right in front of it -- shared by the declarations that follow it without another
    comment in between -- or the one on its own line."""
    src = open(os.path.join(ROOT, "include", "jefferson.h")).read()
    for cite in ("Audio.cu:94", "Audio.cu:164-175", "SoundSource.cu:20-36", "SoundSource.cu:41-54",
                 "hrtf_signals.cu:107-153", "GPUSoundSource.cu:463-471", "cudaPart.cu:21-63"):
        assert cite in src, cite
    # walk the header: comments and the code between them
    parts = re.split(r"(/\*.*?\*/)", src, flags=re.S)
    last_block_comment, uncited, seen = "", [], 0
    for i, part in enumerate(parts):
        if part.startswith("/*"):
            before = parts[i - 1] if i else ""
            own_line = before == "" or before.rstrip(" \t").endswith("\n") or before.strip() == ""
            if own_line:
                last_block_comment = part
            continue
        trailing = parts[i + 1] if i + 1 < len(parts) and parts[i + 1].startswith("/*") else ""
        decls = list(re.finditer(r"\b(jf_[a-z0-9_]+)\s*\(", part))
        for k, m in enumerate(decls):
            seen += 1
            # a trailing comment belongs to the last declaration in front of it
            own = trailing if k == len(decls) - 1 and not part[m.end():].count(";\n") else ""
            if not (_CITE.search(last_block_comment) or _CITE.search(own)):
                uncited.append(m.group(1))
    assert seen >= 35
    assert not uncited, uncited


def test_no_gpu_means_loud_failure(jf, hrir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(jf.JfError) as ei:
        jf.Engine(256, 512, 1, hrir=hrir)
    assert ei.value.code == jf.JF_ERR_DEVICE


def test_create_argument_checks(jf, hrir):
    cfg_bad = [(100, 512, 1, 1), (256, 512, 0, 1), (256, 100, 1, 1), (256, 512, 1, 0), (512, 512, 1, 1)]
    for B, L, S, K in cfg_bad:
        with pytest.raises(jf.JfError) as ei:
            jf.Engine(B, L, S, hrir=hrir, max_batch_blocks=K)
        assert ei.value.code == jf.JF_ERR_ARG, (B, L, S, K)
    with pytest.raises(jf.JfError) as ei:
        jf.Engine(256, 512, 1, hrir_dir="/nonexistent/kemar")
    assert ei.value.code == jf.JF_ERR_IO


def test_product_code_does_not_touch_the_oracle():
    """The shipped library and package must not link, import or call oracle/."""
    pkg = os.path.join(ROOT, "jefferson-2.0_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "jf_oracle" not in text and "model64" not in text and "jfo_" not in text, f
    out = os.popen(f"ldd '{os.path.join(pkg, 'libjefferson_hip.so')}'").read()
    assert "jf_oracle" not in out


def test_host_interpolation_matches_oracle(jf):
    for ele in range(-52, 94):
        for azi in list(range(-3, 364, 5)) + [359, 360, 361]:
            a, b = jf.interpolation(float(ele), float(azi)), oracle_lib.interp(float(ele), float(azi))
            assert (a is None) == (b is None), (ele, azi)
            if a is not None:
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (ele, azi)
    for ele in (-40, 0, 37, 90):
        for azi in range(0, 361):
            assert jf.pick_hrtf(ele, azi) == oracle_lib.pick_hrtf(ele, azi)


def test_host_corrected_interpolation_matches_oracle(jf):
    """jf_interpolation_ex(JF_FLAG_CORRECTED_INTERPOLATION): the host twin of the kernel rule, bit for bit."""
    for ele in np.arange(-45, 91, 2.5):
        for azi in np.arange(-10, 371, 3.1):
            a = jf.interpolation(float(ele), float(azi), jf.JF_FLAG_CORRECTED_INTERPOLATION)
            b = oracle_lib.interp(float(ele), float(azi), corrected=True)
            assert (a is None) == (b is None)
            if a is not None:
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert jf.interpolation(5.0, 3.0, 0)[0].tolist() == jf.interpolation(5.0, 3.0)[0].tolist()


def test_host_positions_match_oracle(jf):
    rng = np.random.default_rng(5)
    for _ in range(200):
        ele, azi, r = rng.uniform(-49, 90), rng.uniform(0, 360), rng.uniform(0.1, 5)
        assert np.array_equal(jf.position_from_spherical(ele, azi, r), oracle_lib.from_spherical(ele, azi, r))
        x, y, z = rng.uniform(-3, 3, 3)
        p, o = jf.position_from_cartesian(x, y, z), oracle_lib.from_cartesian(x, y, z)
        assert np.array_equal(p[:2], o[:2]) and np.array_equal(p[2:], np.float32([x, y, z]))
    assert jf.position_from_cartesian(0, 0, 0) is None
    ele = rng.integers(-40, 90, (7, 5)).astype(np.float32)
    azi = rng.integers(0, 360, (7, 5)).astype(np.float32)
    v = jf.positions_from_spherical(ele, azi, np.float32(1.5))
    assert v.shape == (7, 5, 5)
    assert np.array_equal(v[3, 2], jf.position_from_spherical(ele[3, 2], azi[3, 2], 1.5))


def _write_wav(path, data, sampwidth, nch=1, fs=44100):
    with wave.open(path, "wb") as w:
        w.setnchannels(nch)
        w.setsampwidth(sampwidth)
        w.setframerate(fs)
        w.writeframes(data)


def test_wav_reader_scaling_and_stereo_downmix(jf, tmp_path):
    """cudaPart.cu:21-63 readFile with libsndfile's float scaling."""
    ints = np.array([0, 1, -1, 32767, -32768, 12345], np.int16)
    p = str(tmp_path / "m16.wav")
    _write_wav(p, ints.tobytes(), 2)
    x, fs = jf.wav_read_mono(p)
    assert fs == 44100 and np.array_equal(x, ints.astype(np.float32) / np.float32(32768))

    v24 = np.array([0, 1, -1, 8388607, -8388608, 1234567], np.int32)
    raw = b"".join(struct.pack("<i", int(v))[:3] for v in v24)
    p = str(tmp_path / "m24.wav")
    _write_wav(p, raw, 3)
    x, _ = jf.wav_read_mono(p)
    assert np.array_equal(x, (v24 / 8388608.0).astype(np.float32))

    st = np.array([[1000, 3000], [-2000, 2000], [32767, 32767]], np.int16)
    p = str(tmp_path / "s16.wav")
    _write_wav(p, st.tobytes(), 2, nch=2)
    x, _ = jf.wav_read_mono(p)
    f = st.astype(np.float32) / np.float32(32768)
    assert np.array_equal(x, (f[:, 0] / 2.0 + f[:, 1] / 2.0).astype(np.float32))

    with pytest.raises(jf.JfError):
        jf.wav_read_mono(str(tmp_path / "missing.wav"))
    (tmp_path / "junk.wav").write_bytes(b"not a wav file at all")
    with pytest.raises(jf.JfError):
        jf.wav_read_mono(str(tmp_path / "junk.wav"))


def test_wav_reader_survives_malformed_files(jf, tmp_path):
    """The loader is on the drop-in boundary (cudaPart.cu:21-63 took any path from the command line): random
    corruptions and truncations of valid files must end in an error code or a bounded result, never a crash,
    an exception through the C ABI or an allocation sized by a lying header."""
    rng = np.random.default_rng(7)
    base = []
    for sw, nch in ((2, 1), (3, 2), (1, 1)):
        q = str(tmp_path / f"b{sw}{nch}.wav")
        _write_wav(q, rng.integers(0, 256, 600 * sw * nch, dtype=np.uint8).tobytes(), sw, nch=nch)
        base.append(open(q, "rb").read())
    p = str(tmp_path / "fuzz.wav")
    outcomes = {"ok": 0, "err": 0}
    for trial in range(400):
        b = bytearray(base[trial % len(base)])
        kind = trial % 4
        if kind == 0:      # flip a few header bytes
            for _ in range(3):
                b[rng.integers(0, 44)] = rng.integers(0, 256)
        elif kind == 1:    # truncate anywhere
            b = b[: rng.integers(0, len(b))]
        elif kind == 2:    # sizes that lie (4 GB data chunk, huge fmt chunk)
            off = 40 if rng.integers(0, 2) else 16
            b[off:off + 4] = struct.pack("<I", int(rng.integers(2 ** 31, 2 ** 32)))
        else:              # random bytes after a valid RIFF/WAVE tag
            b = bytearray(b[:12]) + bytearray(rng.integers(0, 256, 200, dtype=np.uint8).tobytes())
        with open(p, "wb") as f:
            f.write(bytes(b))
        try:
            x, _ = jf.wav_read_mono(p)
            assert x.size <= len(b)            # never more samples than bytes present
            outcomes["ok"] += 1
        except jf.JfError:
            outcomes["err"] += 1
    assert outcomes["ok"] > 20 and outcomes["err"] > 20


def test_wav_reader_on_the_castanets_fixture(jf, castanets, tmp_path):
    ex = np.load(os.path.join(GOLD, "castanets_441_excerpt_i24.npy"))
    raw = b"".join(struct.pack("<i", int(v))[:3] for v in ex[:5000])
    p = str(tmp_path / "c.wav")
    _write_wav(p, raw, 3)
    x, _ = jf.wav_read_mono(p)
    assert np.array_equal(x, castanets[:5000])


def test_wav_writer_pcm24_roundtrip(jf, tmp_path):
    rng = np.random.default_rng(2)
    y = rng.uniform(-1, 1, (300, 2)).astype(np.float32)
    y[0] = (1.5, -1.5)  # clipped
    p = str(tmp_path / "o.wav")
    jf.wav_write_stereo24(p, y)
    with wave.open(p) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (2, 3, 44100, 300)
        raw = np.frombuffer(w.readframes(300), np.uint8).reshape(-1, 3).astype(np.int32)
    v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
    v = np.where(v >= 1 << 23, v - (1 << 24), v).reshape(300, 2)
    assert v[0, 0] == 8388607 and v[0, 1] == -8388608
    # float32 scaling (as libsndfile) then rounding: within one 24-bit step
    assert np.abs(v[1:] / 8388607.0 - y[1:]).max() <= 1.0 / 8388607


@pytest.mark.skipif(not os.path.isdir("/root/reference/Jefferson/compact"),
                    reason="the reference's compact KEMAR directory only exists in the build container")
def test_compact_directory_loader_reproduces_the_fixture(jf, hrir):
    """jf_engine_create_from_dir reads the reference's own data files: up to the point where a
    GPU is needed, i.e. a complete, valid directory yields JF_ERR_DEVICE (not JF_ERR_IO)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("checked on the GPU by test_gpu_engine.py")
    with pytest.raises(jf.JfError) as ei:
        jf.Engine(256, 512, 1, hrir_dir="/root/reference/Jefferson/compact")
    assert ei.value.code == jf.JF_ERR_DEVICE


def test_reverb_rms_gain_rule(jf):
    """cudaPart.cu:118,161-186: rms(x) / rms(circular convolution of length n + ceil(n_ir/2))."""
    rng = np.random.default_rng(4)
    for n, n_ir in [(1000, 64), (777, 301), (4096, 1000)]:
        x = rng.uniform(-.5, .5, n).astype(np.float32)
        h = (rng.standard_normal(n_ir) * np.exp(-3 * np.arange(n_ir) / n_ir)).astype(np.float32)
        new_size = n + (n_ir - n_ir // 2)
        X = np.fft.fft(np.pad(x.astype(np.float64), (0, new_size - n)))
        H = np.fft.fft(np.pad(h.astype(np.float64), (0, new_size - n_ir)))
        y = np.fft.ifft(X * H).real
        want = np.sqrt((x.astype(np.float64) ** 2).sum() / (y ** 2).sum())
        assert jf.reverb_rms_gain(x, h) == pytest.approx(want, rel=1e-6)
    assert jf.reverb_rms_gain(np.zeros(10, np.float32), np.ones(3, np.float32)) == 1.0


def test_workload_helpers(jf):
    import importlib.util
    spec = importlib.util.spec_from_file_location("wl", os.path.join(ROOT, "jefferson-2.0_amd", "workload.py"))
    wl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wl)
    assert [wl.shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert wl.shard_range(8192, 8, 7) == (7168, 8192)
    sig, ele, azi, r = wl.source_signal_and_start(5)
    assert len(sig) == 44100 and abs(sig).max() <= 0.5 and (ele, azi) == (-5, 185) and 0.5 <= r <= 3.5
    assert wl.source_signal_and_start(5, 10)[3] == r
    pos = wl.trajectories(jf, [0, 5, 1029], 4)
    assert pos.shape == (4, 3, 5)
    assert pos[:, 1, 1].tolist() == [185, 186, 187, 188] and set(pos[:, 1, 0]) == {-5.0}
    # shards of a multi-GPU job are slices of the single-GPU job
    assert np.array_equal(wl.trajectories(jf, [1029], 4)[:, 0], pos[:, 2])
    terms = wl.n_terms_table(jf)
    assert terms[0 + 49, 0] == 1 and terms[0 + 49, 3] == 2 and terms[5 + 49, 0] == 2 and terms[5 + 49, 3] == 4
    # stationary case-4 source: 4 rows + window + output = 38 976 B
    one = np.tile(jf.position_from_spherical(5, 3, 1.0), (3, 1, 1))
    b, rows, items = wl.algorithmic_bytes(jf, one, 256, first_old=np.array([[5, 3]]), terms=terms)
    assert (b, rows, items) == (3 * 38976, 12, 3)
    mov = wl.trajectories(jf, [2], 3, first_block=2)  # ele -26, azi 76, 77, 78: both ends case 4 -> 71 808 B
    b, rows, items = wl.algorithmic_bytes(jf, mov[1:], 256, first_old=mov[0, :, :2].astype(np.int64), terms=terms)
    assert b == 2 * 71808
    # bench.py walks one uploaded period of the trajectory cyclically: same bytes as the unrolled trajectory
    period = wl.trajectories(jf, [0, 5, 77], 360)                     # azimuth + 1 degree per block: 360 blocks
    long = wl.trajectories(jf, [0, 5, 77], 360 + 3 * 360)
    assert np.array_equal(long[360:720], period)
    first, n = 7, 3 * 360 // 8 - 5                                      # steps of 8 blocks, well past one period
    cyc = wl.algorithmic_bytes_cyclic(jf, period, 8, 256, first, n, terms=terms)
    direct = wl.algorithmic_bytes(jf, long[8 * first:8 * (first + n)], 256,
                                  first_old=long[8 * first - 1, :, :2].astype(np.int64), terms=terms)
    assert cyc == direct


def test_flop_model(jf):
    """The fp32 work model behind bench.py's roofline (workload.flops_window), case by case."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("wl", os.path.join(ROOT, "jefferson-2.0_amd", "workload.py"))
    wl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wl)
    terms = wl.n_terms_table(jf)
    B = 256
    front = wl.FLOPS_RFFT + wl.FLOPS_DISTANCE
    assert wl.FLOPS_RFFT == 23040 + 6144 and wl.FLOPS_IFFT_FULL == 51200
    inv = wl.flops_ifft_pruned(B)
    assert inv == 40960 + 6144
    # one stationary case-4 source, 3 blocks, per-source kernel: front + 4-row filter + one pruned inverse + mix add
    one = np.tile(jf.position_from_spherical(5, 3, 1.0), (3, 1, 1))
    ex, ref = wl.flops_window(jf, one, B, 1, first_old=np.array([[5, 3]]), terms=terms)
    assert ex == 3 * (front + wl.flops_filter(4) + inv + 2 * B)
    assert ref == 3 * (front + wl.flops_filter(4) + wl.FLOPS_IFFT_FULL + 2 * B)
    # two moving case-4 sources in one unit (G = 2): two filters per source, ONE inverse per set per unit
    mov = wl.trajectories(jf, [2, 1026], 3, first_block=2)
    e_, a_ = mov[..., 0].astype(int), mov[..., 1].astype(int)
    n_new = terms[e_[1:] + 49, a_[1:]]
    n_old = terms[e_[:-1] + 49, a_[:-1]]          # every block moves: the old set is the previous block's
    assert n_new[:, 0].tolist() == [4, 4]         # source 2: ele -26, azimuths 77, 78 -> case 4 at both ends
    ex2, ref2 = wl.flops_window(jf, mov[1:], B, 2, first_old=mov[0, :, :2].astype(np.int64), terms=terms)
    filt = sum(wl.flops_filter(int(n)) for n in n_new.ravel()) + sum(wl.flops_filter(int(n)) for n in n_old.ravel())
    per_unit = 2 * inv + B * wl.FLOPS_XFADE_PER_FRAME + 2 * B
    assert ex2 == 4 * (front + wl.NC * 4 * 2) + filt + 2 * per_unit
    assert ref2 == 4 * (front + 2 * wl.FLOPS_IFFT_FULL + B * wl.FLOPS_XFADE_PER_FRAME + 2 * B) + filt
    # the round-1 group kernel inverted the old sets per source
    ex3, _ = wl.flops_window(jf, mov[1:], B, 2, first_old=mov[0, :, :2].astype(np.int64), terms=terms,
                             old_sets_spectral=False)
    assert ex3 == ex2 + 2 * inv
    # cyclic walk == unrolled trajectory
    period = wl.trajectories(jf, [0, 5, 77, 400], 360)
    long = wl.trajectories(jf, [0, 5, 77, 400], 4 * 360)
    first, n = 7, 100
    cyc = wl.flops_cyclic(jf, period, 8, B, 4, first, n, terms=terms)
    direct = wl.flops_window(jf, long[8 * first:8 * (first + n)], B, 4,
                             first_old=long[8 * first - 1, :, :2].astype(np.int64), terms=terms)
    assert cyc == direct


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher spawns torch.distributed.run itself and hands back the
    children's failure (here: no GPU) as a non-zero exit code -- it neither refuses to start nor hangs."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = r.stderr.decode(errors="replace")
    assert r.returncode != 0
    assert "launch with torch.distributed.run" not in err
    assert "needs a GPU" in err          # printed by the ranks, i.e. they were started


def test_bench_also_legs_and_their_briefs():
    """bench.py's secondary legs (round 6: config 5 in batch form and in real time, the stationary variant, under "also" of the
    default line): configured from the headline's arguments without touching them; the record keeps what the brief names."""
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("jf_bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    a = argparse.Namespace(steps=20, warmup=5, reverb=False, stationary=False, realtime=False, move_every=1, rv_sources=256,
                           rv_ir_seconds=2.0, pmc_child=False, no_pmc=False)
    legs = b.also_configurations(a)
    assert list(legs) == ["reverb", "reverb_realtime_us", "stationary"]
    assert legs["reverb"].reverb and legs["reverb"].steps >= 32 and legs["reverb"].warmup >= 8 and legs["reverb"].also_leg
    assert legs["stationary"].stationary and legs["stationary"].steps >= 20 and not legs["stationary"].reverb
    assert not a.reverb and not a.stationary and a.steps == 20 and not hasattr(a, "also_leg")   # the headline's own arguments
    assert b.leg_pmc_kernels(legs["reverb"]) == ("reverb_big_mac_kernel", "reverb_mac_tiled_kernel")
    assert b.precollect_pmc(a, 2)[0] is None and "N = 1" in b.precollect_pmc(a, 2)[1]
    line = {"metric": "m", "value": 1.0, "unit": "u", "steps": 32, "warmup": 8, "prewarm_steps": 248, "ms_per_step": 0.3,
            "real_time_factor": 2.0, "verified": True, "step_split_ms": {"reverb_stage": 0.1},
            "config": {"workload": "w", "sources_per_gpu": 256, "block": 128, "blocks_per_step": 256, "source_group": 16,
                       "kernels": ["k"], "interp_table": {}, "parallelism": "1 GPU"},
            "roofline": {"kernel": "k", "bound": "hbm", "avg_stage_ms": 0.11, "algorithmic_bytes_per_step": 5e8, "achieved": 4.6e3,
                         "peak": 8e3, "unit": "GB/s", "frac": 0.58, "traffic": 3.5e8, "flops_per_step": 1.0},
            "cpu_baseline": {"value": 3e7, "unit": "u", "cores": 16, "kind": "port", "sample": "s", "gpu_over_cpu": 1e3},
            "verification": {"max_abs_err_mix": 1e-7, "mix_peak": 1.0, "bound_mix": 1e-5, "against": "oracle"}}
    br = b.brief(line, ("kernel", "bound", "avg_stage_ms", "algorithmic_bytes_per_step", "frac", "traffic"))
    assert br["verified"] is True and br["roofline"] == {"kernel": "k", "bound": "hbm", "avg_stage_ms": 0.11,
                                                         "algorithmic_bytes_per_step": 5e8, "frac": 0.58, "traffic": 3.5e8}
    assert "parallelism" not in br["config"] and br["cpu_baseline"]["cores"] == 16 and b.brief(None, ()) is None


# ---------------------------------------------------------------- jefferson_group.h (one job over a node's GPUs) --
def _group(jf):
    import importlib
    return importlib.import_module("jefferson_amd.group")


def test_group_library_exports_and_header(jf):
    """libjefferson_group.so loads next to libjefferson_hip.so and exports every entry point its header declares."""
    import ctypes
    import re
    grp = _group(jf)
    L = ctypes.CDLL(grp.LIB_PATH)
    text = open(grp.HEADER_PATH).read()
    text_dbg = open(os.path.join(ROOT, "include", "jefferson_debug.h")).read()   # its test support is declared there
    declared = set(re.findall(r"\b(jf_(?:group_\w+|shard_range))\s*\(", _strip_comments(text) + _strip_comments(text_dbg)))
    assert not re.findall(r"\bjf_group_(?:debug_\w+|create_shards_on_device)\s*\(", _strip_comments(text))
    assert declared == set(grp.exported_symbols())
    for name in declared:
        assert hasattr(L, name), name
    assert "Audio.cu:109-110" in text      # the interface it replaces is cited


def test_group_shard_range_is_the_bench_partition(jf):
    """jf_shard_range (C) == workload.shard_range (bench.py / tests): contiguous, covering, sizes within one."""
    import importlib.util
    grp = _group(jf)
    spec = importlib.util.spec_from_file_location("wl", os.path.join(ROOT, "jefferson-2.0_amd", "workload.py"))
    wl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wl)
    for n_total in (1, 5, 8, 1000, 1024, 8192, 8191):
        for parts in (1, 2, 3, 7, 8):
            got = [grp.shard_range(n_total, parts, r) for r in range(parts)]
            assert got == [wl.shard_range(n_total, parts, r) for r in range(parts)]
            assert got[0][0] == 0 and got[-1][1] == n_total
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            sizes = [hi - lo for lo, hi in got]
            assert max(sizes) - min(sizes) <= 1
    assert grp.shard_range(10, 0, 0) is None and grp.shard_range(10, 2, 2) is None and grp.shard_range(10, 2, -1) is None


def test_group_without_gpu_fails_loudly(jf, hrir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    grp = _group(jf)
    with pytest.raises(jf.JfError) as ei:
        grp.Group(256, 512, 4, hrir, n_gpus=1)
    assert ei.value.code == jf.JF_ERR_DEVICE
    with pytest.raises(jf.JfError) as ei:
        grp.Group(256, 512, 2, hrir, n_gpus=3)      # more GPUs than sources
    assert ei.value.code == jf.JF_ERR_ARG


def test_every_entry_point_survives_null_arguments():
    """No exit, no throw and no fault across the ABI (the reference prints and exits, cufftDefines.cuh:69-77): every
    exported function of both libraries called with a null engine / group and null or zero everything else must come back --
    with an error code where it has one.  In a child process, each name printed before its call, so that a fault names its
    function.  No GPU needed (and none touched: a null handle is refused first)."""
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
from jf_load import jf
import importlib
mods = [(jf, jf.lib(), jf._SIGS)]
try:
    grp = importlib.import_module("jefferson_amd.group")
    mods.append((grp, grp.lib(), grp._SIGS))
except Exception as ex:      # the group library needs RCCL at build time
    print("no group library:", ex)
def zero(t):
    if t in (C.c_int, C.c_uint, C.c_long, C.c_longlong, C.c_size_t, C.c_ulong):
        return t(0)
    if t in (C.c_float, C.c_double):
        return t(0.0)
    return None            # pointers of every kind
n = 0
for mod, L, sigs in mods:
    for name, (res, args) in sorted(sigs.items()):
        print("calling", name, flush=True)
        r = getattr(L, name)(*[zero(t) for t in args])
        n += 1
print("SURVIVED", n)
''' % ROOT
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "SURVIVED" in out, (out[-600:], r.stderr.decode()[-600:])
    assert int(out.split("SURVIVED")[1]) >= 80
