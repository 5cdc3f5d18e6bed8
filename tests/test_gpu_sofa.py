"""jf_engine_create_sofa on a real MI355X (SOFA / other HRTF sets):

* the four h5py-written containers of tests/golden/sofa (one set: 33 measurements on five rings) -- an engine created from the
  FILE renders what an engine created from the set's table and rings renders, bit for bit, and that is the float32 C oracle's
  and the float64 model's output for the arrays the fixture generator wrote (which never went through the reader);
* KEMAR itself written as a SOFA file by h5py in a child process (when the image's second interpreter is there): the 710 rows
  land where the reference's loader puts them (the spectra table bit for bit);
* the offline driver takes a .sofa name where it takes the KEMAR directory.
"""
import os
import struct
import subprocess
import wave

import numpy as np
import pytest

import model64
import oracle_lib
from conftest import ROOT, assert_within, scenario_positions, sum_tol

pytestmark = pytest.mark.gpu

SOFA = os.path.join(ROOT, "tests", "golden", "sofa")
CONDA_PY = "/opt/conda/bin/python3.9"
TOL64 = 2e-7
TOL32 = 4e-7


def _trajectory(jf, S, K):
    pos = np.zeros((K, S, 5), np.float32)
    for s in range(S):
        for k in range(K):
            ele = -40 + (17 * s + (9 * k if s % 2 else 0)) % 131
            azi = (47 * s + (k if s % 3 == 0 else 40 * k)) % 360
            pos[k, s] = jf.position_from_spherical(float(ele), float(azi), 0.3 + 0.05 * s)
            pos[k, s, 0] = ele
    return pos


@pytest.mark.parametrize("name", ["nc4", "symtab", "latest", "cartesian"])
def test_engine_from_a_sofa_file(jf, castanets, name):
    E = np.load(os.path.join(SOFA, "sofa_expected.npz"))
    path = os.path.join(SOFA, name + ".sofa")
    S, K, B = 6, 8, 256
    pos = _trajectory(jf, S, K)
    sigs = [(0.45 * np.roll(castanets, 2003 * s)[:9000 + 97 * s]).astype(np.float32) for s in range(S)]

    def run(e):
        assert e.table_rows() == 33
        for s in range(S):
            e.set_signal(s, sigs[s])
        a = e.process_batch(pos[:6])
        got = [a]
        for k in (6, 7):
            e.set_latched(pos[k])
            got.append(e.process_block()[None])
        e.close()
        return np.concatenate(got)

    from_file = run(jf.Engine(B, 512, S, sofa=path, sofa_tol_deg=0.01, max_batch_blocks=6))
    st = jf.SofaSet(path)
    grid, hrir = st.table(tol_deg=0.01)
    st.close()
    from_table = run(jf.Engine(B, 512, S, hrir=hrir, grid=grid, max_batch_blocks=6))
    assert np.abs(from_file).max() > 0.02 and np.array_equal(from_file, from_table)

    # the oracles on the generator's own arrays (ring order; the per-measurement delays of cartesian.sofa applied here)
    delay = E["delay"] if name == "cartesian" else np.zeros((33, 2), np.int64)
    h = np.zeros((33, 2, 24 + int(delay.max())), np.float32)
    for row in range(33):
        for ear in range(2):
            d = int(delay[row, ear])          # (E["delay"] is in the set's row order: the generator wrote delay[file order])
            h[row, ear, d:d + 24] = E["ir"][row, ear]
    assert np.array_equal(h, hrir)
    og = oracle_lib.Grid(E["ring_ele"].tolist(), E["ring_count"].tolist())
    mg = model64.Grid(E["ring_ele"].tolist(), E["ring_count"].tolist())
    ora = oracle_lib.Engine(B, 512, S, h, grid=og)
    mod = model64.Model(B, 512, S, h, grid=mg)
    for s in range(S):
        ora.set_signal(s, sigs[s])
        mod.set_signal(s, sigs[s])
    want32, _ = ora.process_batch(pos, want_partial=True)
    want64, _ = mod.process_batch(pos)
    ora.close()
    assert_within(from_file, want64, sum_tol(TOL64, S), f"sofa {name}: vs model64")
    assert_within(from_file, want32, sum_tol(TOL32, S), f"sofa {name}: vs oracle32")


def test_group_from_a_sofa_file(jf, castanets):
    """jf_group_create_sofa (one GPU: a communicator of size 1) renders what jf_engine_create_sofa renders"""
    import importlib
    grp = importlib.import_module("jefferson_amd.group")       # needs libjefferson_group.so (RCCL at build time)
    path = os.path.join(SOFA, "nc4.sofa")
    S, K, B = 6, 6, 128
    pos = _trajectory(jf, S, K)
    sigs = [(0.45 * np.roll(castanets, 2003 * s)[:9000 + 97 * s]).astype(np.float32) for s in range(S)]
    e = jf.Engine(B, 512, S, sofa=path, sofa_tol_deg=0.01, max_batch_blocks=K)
    G = grp.Group(B, 512, S, None, n_gpus=1, max_batch_blocks=K, sofa=path, sofa_tol_deg=0.01)
    for s in range(S):
        e.set_signal(s, sigs[s])
        G.set_signal(s, sigs[s])
    want, got = e.process_batch(pos), G.process_batch(pos)
    e.close()
    G.close()
    assert np.abs(want).max() > 0.02 and np.array_equal(got, want)
    with pytest.raises(jf.JfError) as ex:
        grp.Group(B, 512, S, None, n_gpus=1, sofa=os.path.join(SOFA, "mono.sofa"))
    assert ex.value.code == jf.JF_ERR_IO and "receivers" in str(ex.value)


def test_hrtf_len_must_hold_the_sets_taps(jf):
    with pytest.raises(jf.JfError) as ex:
        jf.Engine(256, 16, 1, sofa=os.path.join(SOFA, "nc4.sofa"))      # 24 taps
    assert ex.value.code == jf.JF_ERR_ARG
    with pytest.raises(jf.JfError) as ex:
        jf.Engine(256, 512, 1, sofa=os.path.join(SOFA, "mono.sofa"))
    assert ex.value.code == jf.JF_ERR_IO and "receivers" in str(ex.value)


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="the image's interpreter with h5py is not there")
def test_kemar_as_a_sofa_file(jf, hrir, castanets, tmp_path):
    """KEMAR's 710 impulse responses written by libhdf5 as a SimpleFreeFieldHRIR file -- shuffled, azimuths the whole degrees
    of the files' names in SOFA's counter-clockwise sense, deflated float64 -- and read back by the library: the table's
    rows are the reference loader's rows (spectra bit for bit), and the engine is jf_engine_create's engine."""
    pos = model64.table_positions()
    np.save(tmp_path / "hrir.npy", hrir)
    np.save(tmp_path / "ele.npy", np.array([e for e, _ in pos], np.float64))
    np.save(tmp_path / "azi.npy", np.array([a for _, a in pos], np.float64))
    script = tmp_path / "write_kemar.py"
    script.write_text('''
import sys, warnings
warnings.simplefilter("ignore")
import h5py, numpy as np
d = sys.argv[1]
hrir, ele, azi = np.load(d + "/hrir.npy"), np.load(d + "/ele.npy"), np.load(d + "/azi.npy")
order = np.random.default_rng(5).permutation(len(ele))
with h5py.File(d + "/kemar.sofa", "w", libver="earliest", track_order=True) as f:
    for k, v in (("Conventions", "SOFA"), ("SOFAConventions", "SimpleFreeFieldHRIR"), ("DataType", "FIR")):
        f.attrs.create(k, np.bytes_(v.encode() + b"\\0"))
    f.create_dataset("Data.IR", data=hrir[order].astype(np.float64), chunks=(64, 2, 128), compression="gzip", shuffle=True)
    p = f.create_dataset("SourcePosition", data=np.stack([(360.0 - azi[order]) % 360.0, ele[order], np.full(len(ele), 1.4)], axis=1))
    p.attrs.create("Type", np.bytes_(b"spherical\\0"))
    p.attrs.create("Units", np.bytes_(b"degree, degree, metre\\0"))
    f.create_dataset("Data.SamplingRate", data=np.array([44100.0]))
    f.create_dataset("Data.Delay", data=np.zeros((1, 2)))
np.save(d + "/order.npy", order)
''')
    r = subprocess.run([CONDA_PY, str(script), str(tmp_path)], capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        pytest.skip("h5py is not usable here: " + r.stderr[-300:])
    path = str(tmp_path / "kemar.sofa")
    st = jf.SofaSet(path)
    grid, table = st.table(tol_deg=0.51)
    st.close()
    assert grid.count.tolist() == [56, 60, 72, 72, 72, 72, 72, 60, 56, 45, 36, 24, 12, 1] and np.array_equal(table, hrir)
    e = jf.Engine(256, 512, 4, sofa=path, sofa_tol_deg=0.51, max_batch_blocks=6)
    ref = jf.Engine(256, 512, 4, hrir=hrir, max_batch_blocks=6)
    assert e.table_rows() == 710 and np.array_equal(e.read_table(), ref.read_table())
    ref.close()
    # KEMAR's rings are recognised: the reference's description (its rounded steps), the reference's rule -- the engine from the
    # SOFA file IS jf_engine_create (and, with the flag, the corrected rule's engine), FD_BASIC included
    assert np.array_equal(grid.step, jf.Grid.kemar().step)
    tr = _trajectory(jf, 4, 6)
    tr[:, :, 0] = np.maximum(tr[:, :, 0], -40)            # (the reference's range)
    for flags in (0, jf.JF_FLAG_CORRECTED_INTERPOLATION):
        out = []
        for eng in (e if flags == 0 else jf.Engine(256, 512, 4, sofa=path, sofa_tol_deg=0.51, max_batch_blocks=6, flags=flags),
                    jf.Engine(256, 512, 4, hrir=hrir, max_batch_blocks=6, flags=flags)):
            for s in range(4):
                eng.set_signal(s, (0.4 * np.roll(castanets, 1500 * s)[:8000]).astype(np.float32))
            a = eng.process_batch(tr[:4])
            eng.set_mode(jf.JF_MODE_FD_BASIC)
            out.append(np.concatenate([a, eng.process_batch(tr[4:])]))
            assert eng.set_spherical(0, -60.0, 0.0, 1.0) == jf.JF_ERR_RANGE
            eng.close()
        assert np.abs(out[0]).max() > 0.02 and np.array_equal(out[0], out[1])


def test_offline_driver_on_a_sofa_file(jf, tmp_path):
    ex = np.load(os.path.join(ROOT, "tests", "golden", "castanets_441_excerpt_i24.npy"))[:20000]
    inp, outp = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    with wave.open(inp, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(3)
        w.setframerate(44100)
        w.writeframes(b"".join(struct.pack("<i", int(v))[:3] for v in ex))
    exe = os.path.join(ROOT, "jefferson-2.0_amd", "jf_render")
    sofa = os.path.join(SOFA, "latest.sofa")
    r = subprocess.run([exe, sofa, inp, outp, "--block", "256", "--azi", "3", "--ele", "5", "--dwell", "3", "--rounds", "4",
                        "--sofa-tol", "0.01"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    with wave.open(outp) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getnframes()) == (2, 3, 256 * 3 * 5)
        raw = np.frombuffer(w.readframes(w.getnframes()), np.uint8).reshape(-1, 3).astype(np.int32)
    v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
    got = (np.where(v >= 1 << 23, v - (1 << 24), v) / 8388607.0).reshape(-1, 512)
    sig, _ = jf.wav_read_mono(inp)
    e = jf.Engine(256, 512, 1, sofa=sofa, sofa_tol_deg=0.01)
    e.set_signal(0, sig)
    want = []
    for (ele, azi, rr) in scenario_positions(3, 5, 3, 4):
        assert e.set_spherical(0, ele, azi, rr) == 0
        want.append(e.process_block())
    e.close()
    assert np.abs(np.array(want)).max() > 0.01
    assert np.abs(got - np.array(want)).max() <= 1.0 / 8388607
    # a file that is no set: the driver says why and fails
    r = subprocess.run([exe, os.path.join(SOFA, "mono.sofa"), inp, outp], capture_output=True, text=True)
    assert r.returncode == 1 and "receivers" in r.stderr
