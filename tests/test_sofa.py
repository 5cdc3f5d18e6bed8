"""HRTF sets in SOFA files (include/jefferson.h: jf_sofa_*; SURVEY.md 8(f)-2 "SOFA/other HRTF sets", the reference's TODO
FuturePlans.md:21) -- host side, no GPU.

The library reads the HDF5 container itself (csrc/jf_hdf5.c).  What pins that reader:
  * tests/golden/sofa/*.sofa: one set in four containers WRITTEN BY libhdf5 1.10.6 (h5py; golden/make_sofa_fixtures.py says which
    HDF5 structures each one exercises) against the arrays the generator wrote beside them, bit for bit;
  * when the image's second interpreter is there (/opt/conda/bin/python3.9 with h5py): every numeric dataset of the HDF5 files
    that ship with PyTables and SciPy in this image (written by other HDF5 versions and by MATLAB) against h5py's reading;
  * damaged files: truncations and byte flips yield errors, never faults.
"""
import json
import os
import subprocess

import numpy as np
import pytest

from jf_load import jf

HERE = os.path.dirname(os.path.abspath(__file__))
SOFA = os.path.join(HERE, "golden", "sofa")
CONDA_PY = "/opt/conda/bin/python3.9"


@pytest.fixture(scope="module")
def expected():
    return np.load(os.path.join(SOFA, "sofa_expected.npz"))


def _order(expected, name):
    key = {"nc4": "order_nc4", "latest": "order_latest", "cartesian": "order_cartesian"}.get(name)
    return expected[key] if key else np.arange(len(expected["el"]))


@pytest.mark.parametrize("name", ["nc4", "symtab", "latest", "cartesian"])
def test_the_containers_libhdf5_wrote(expected, name):
    path = os.path.join(SOFA, name + ".sofa")
    o = _order(expected, name)
    ir = jf.hdf5_read(path, "Data.IR")
    assert ir.shape == (33, 2, 24) and np.array_equal(ir, expected["ir"][o].astype(np.float64))
    pos = jf.hdf5_read(path, "/SourcePosition")
    if name == "cartesian":
        a, e = np.radians(expected["az_sofa"][o]), np.radians(expected["el"][o])
        want = np.stack([2 * np.cos(e) * np.cos(a), 2 * np.cos(e) * np.sin(a), 2 * np.sin(e)], axis=1)
        assert np.array_equal(pos, want)
        assert np.array_equal(jf.hdf5_read(path, "Data.Delay"), expected["delay"][o].astype(np.float64))
        assert jf.hdf5_read(path, "Data.SamplingRate").tolist() == [44100.0]          # an int32 dataset
    else:
        assert np.array_equal(pos, np.stack([expected["az_sofa"][o], expected["el"][o], np.full(33, 1.4)], axis=1))
        assert np.array_equal(jf.hdf5_read(path, "Data.Delay"), np.zeros((1, 2)))
        assert float(jf.hdf5_read(path, "Data.SamplingRate").reshape(-1)[0]) == 44100.0
    assert jf.hdf5_read(path, "ReceiverPosition").shape == (2, 3, 1)
    assert jf.hdf5_attr(path, "/", "SOFAConventions") == "SimpleFreeFieldHRIR"
    assert jf.hdf5_attr(path, "", "DataType") == "FIR"
    assert jf.hdf5_attr(path, "SourcePosition", "Type") == ("cartesian" if name == "cartesian" else "spherical")
    assert jf.hdf5_attr(path, "SourcePosition", "Units") == ("metre" if name == "cartesian" else "degree, degree, metre")
    assert jf.hdf5_attr(path, "/", "NoSuchAttribute") is None
    if name in ("nc4", "latest"):      # 30 global attributes: dense attribute storage
        assert jf.hdf5_attr(path, "/", "Extra3") == "z" and jf.hdf5_attr(path, "/", "DateCreated") == "2026-10-05 00:00:00"
    if name == "latest":
        assert np.array_equal(jf.hdf5_read(path, "paged"), expected["paged"].astype(np.float64))
    if name == "symtab":
        assert jf.hdf5_read(path, "nested/seven").tolist() == [-3, -2, -1, 0, 1, 2, 3]
        assert jf.hdf5_read(path, "/nested/bytes").tolist() == [200, 3]
        assert jf.hdf5_read(path, "nested/wide").tolist() == [-2.0 ** 40, 2.0 ** 40]
        with pytest.raises(jf.JfError) as ex:
            jf.hdf5_read(path, "nested/eight")
        assert ex.value.code == jf.JF_ERR_ARG
        with pytest.raises(jf.JfError) as ex:       # a group is not a dataset
            jf.hdf5_read(path, "nested")
        assert ex.value.code == jf.JF_ERR_IO


@pytest.mark.parametrize("name", ["nc4", "symtab", "latest", "cartesian"])
def test_the_set_and_its_table(expected, name):
    path = os.path.join(SOFA, name + ".sofa")
    o = _order(expected, name)
    s = jf.SofaSet(path)
    assert (s.M, s.R, s.N, s.sample_rate, s.conventions) == (33, 2, 24, 44100.0, "SimpleFreeFieldHRIR")
    assert np.array_equal(s.ir, expected["ir"][o])
    tol = 1e-4 if name == "cartesian" else 0       # (through atan2 in double, then float32)
    assert np.allclose(s.azimuth, expected["az_sofa"][o] % 360, atol=tol) or \
        np.allclose((s.azimuth - expected["az_sofa"][o] + 180) % 360 - 180, 0, atol=tol)
    assert np.allclose(s.elevation, expected["el"][o], atol=tol)
    assert np.allclose(s.distance, 2.0 if name == "cartesian" else 1.4, atol=1e-6)
    delay = expected["delay"][o] if name == "cartesian" else np.zeros((33, 2))
    assert np.array_equal(s.delay, delay.astype(np.float32))
    assert s.taps() == 24 + int(delay.max())
    grid, hrir = s.table(tol_deg=0.01)
    assert grid.ele.tolist() == expected["ring_ele"].tolist() and grid.count.tolist() == expected["ring_count"].tolist()
    assert grid.rows() == 33
    # the table: ring order, azimuth ascending in KEMAR's clockwise sense = the order the generator built the set in;
    # whole-sample delays shift their response
    want = np.zeros((33, 2, s.taps()), np.float32)
    inv = np.argsort(o)                                # measurement inv[row] of the file is the set's row
    for row in range(33):
        for ear in range(2):
            d = int(delay[inv[row], ear])
            want[row, ear, d:d + 24] = expected["ir"][row, ear]
    assert np.array_equal(hrir, want)
    # a wider table pads, a narrower one is refused
    _, wide = s.table(tol_deg=0.01, taps=s.taps() + 5)
    assert np.array_equal(wide[:, :, :s.taps()], want) and not wide[:, :, s.taps():].any()
    with pytest.raises(jf.JfError) as ex:
        s.table(tol_deg=0.01, taps=s.taps() - 1)
    assert ex.value.code == jf.JF_ERR_ARG
    s.close()


def _patched(tmp_path, name, old, new, count=1):
    raw = open(os.path.join(SOFA, name + ".sofa"), "rb").read()
    assert raw.count(old) == count, (name, old, raw.count(old))
    p = tmp_path / (name + "_patched.sofa")
    p.write_bytes(raw.replace(old, new))
    return str(p)


def test_sets_the_engine_cannot_take(tmp_path, expected):
    # one receiver
    s = jf.SofaSet(os.path.join(SOFA, "mono.sofa"))
    assert (s.M, s.R, s.N, s.conventions) == (4, 1, 4, "")
    with pytest.raises(jf.JfError) as ex:
        s.table()
    assert ex.value.code == jf.JF_ERR_IO and "receivers" in str(ex.value)
    # 48 kHz (the contiguous float64 of symtab.sofa's scalar Data.SamplingRate)
    p = _patched(tmp_path, "symtab", np.float64(44100.0).tobytes(), np.float64(48000.0).tobytes())
    s = jf.SofaSet(p)
    assert s.sample_rate == 48000.0
    with pytest.raises(jf.JfError) as ex:
        s.table()
    assert ex.value.code == jf.JF_ERR_IO and "44100" in str(ex.value)
    # transfer functions instead of impulse responses
    p = _patched(tmp_path, "cartesian", b"FIR\0", b"TF\0\0")
    with pytest.raises(jf.JfError) as ex:
        jf.SofaSet(p)
    assert ex.value.code == jf.JF_ERR_IO and "DataType" in str(ex.value)
    # a measurement off its ring's uniform steps (symtab.sofa holds the positions as contiguous float64: row 1 = 315 degrees)
    p = _patched(tmp_path, "symtab", np.array([315.0, -30.0, 1.4]).tobytes(), np.array([300.0, -30.0, 1.4]).tobytes())
    s = jf.SofaSet(p)
    with pytest.raises(jf.JfError) as ex:
        s.table(tol_deg=0.5)
    assert ex.value.code == jf.JF_ERR_ARG
    # a fractional delay (cartesian.sofa's Data.Delay is contiguous float64 [33][2]; measurement 0 of the file)
    o = expected["order_cartesian"]
    row = expected["delay"][o].astype(np.float64)
    p = _patched(tmp_path, "cartesian", row.tobytes(), (row + np.where(np.arange(66).reshape(33, 2) == 5, 0.5, 0.0)).tobytes())
    s = jf.SofaSet(p)
    with pytest.raises(jf.JfError) as ex:
        s.taps()
    assert ex.value.code == jf.JF_ERR_IO and "fractional" in str(ex.value)
    # files that are not HDF5, that do not exist, null arguments
    junk = tmp_path / "junk.sofa"
    junk.write_bytes(b"RIFF" + bytes(5000))
    for path in (str(junk), str(tmp_path / "absent.sofa"), os.path.join(SOFA, "sofa_expected.npz")):
        with pytest.raises(jf.JfError) as ex:
            jf.SofaSet(path)
        assert ex.value.code == jf.JF_ERR_IO
    L = jf.lib()
    assert L.jf_sofa_read(None, None) == jf.JF_ERR_ARG and L.jf_sofa_taps(None) == jf.JF_ERR_ARG
    assert L.jf_sofa_table(None, 0.1, None, None, 4) == jf.JF_ERR_ARG
    L.jf_sofa_release(None)
    # an engine from a file that is not there: the create path reports, no GPU is touched before the file is read
    with pytest.raises(jf.JfError) as ex:
        jf.Engine(256, 512, 1, sofa=str(tmp_path / "absent.sofa"))
    assert ex.value.code == jf.JF_ERR_IO


def test_damaged_files_yield_errors_not_faults(tmp_path):
    """truncations, byte flips, runs of 0xff (undefined addresses) over the four containers: every call returns"""
    rng = np.random.default_rng(11)
    p = str(tmp_path / "damaged.sofa")
    opened = failed = 0
    for name in ("nc4", "symtab", "latest", "cartesian"):
        raw = np.frombuffer(open(os.path.join(SOFA, name + ".sofa"), "rb").read(), np.uint8)
        for it in range(150):
            m = raw.copy()
            kind = it % 4
            if kind == 0:
                m = m[:rng.integers(0, len(m))]
            elif kind == 1:
                m[rng.integers(0, len(m), rng.integers(1, 9))] = rng.integers(0, 256, 1, dtype=np.uint8)
            elif kind == 2:
                at = int(rng.integers(0, len(m) - 8))
                m[at:at + 8] = 0xFF
            else:
                m[rng.integers(0, 4096)] = rng.integers(0, 256, dtype=np.uint8)
            m.tofile(p)
            try:
                s = jf.SofaSet(p)
                s.table(tol_deg=0.5)
                s.close()
                opened += 1
            except jf.JfError as ex:
                assert ex.code in (jf.JF_ERR_IO, jf.JF_ERR_ARG), ex
                failed += 1
            for ds in ("Data.IR", "nested/seven", "paged"):
                try:
                    jf.hdf5_read(p, ds)
                except jf.JfError:
                    pass
    assert opened > 50 and failed > 100      # (many flips land in data, many in structure)


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="the image's interpreter with h5py is not there")
def test_against_h5py_on_files_other_writers_made(tmp_path):
    """Every integer / IEEE float dataset of the HDF5 files that ship with PyTables and SciPy in this image (HDF5 1.6 .. 1.10,
    MATLAB 7.4's; old-style and new-style groups, layout message versions 1-4, chunked + deflate + shuffle ...), read by
    h5py in a child process and by the library here: equal element for element.  Datasets the reader refuses by design
    (enumerations, compounds, strings, ...) are not numeric to h5py either, or are counted."""
    dump = tmp_path / "dump.py"
    dump.write_text('''
import glob, json, sys, warnings
warnings.simplefilter("ignore")
import h5py, numpy as np
out = sys.argv[1]
files = sorted(glob.glob("/opt/conda/lib/python3.9/site-packages/tables/tests/*.h5")
               + glob.glob("/opt/conda/lib/python3.9/site-packages/tables/nodes/tests/*.h5")
               + glob.glob("/usr/local/lib/python3.10/dist-packages/scipy/io/matlab/tests/data/*hdf5*.mat"))
index = {}
for i, fn in enumerate(files):
    found = {}
    try:
        f = h5py.File(fn, "r")
    except Exception:
        continue
    def visit(name, obj):
        if isinstance(obj, h5py.Dataset) and obj.dtype.kind in "iuf" and obj.dtype.itemsize in (1, 2, 4, 8) \\
                and obj.dtype != np.float16 and h5py.check_enum_dtype(obj.dtype) is None:
            try:
                a = obj[()]
            except Exception:
                return
            if a.size < 2000000:
                found[name] = np.asarray(a, dtype=np.float64)
    try:
        f.visititems(visit)
    except Exception:
        pass
    f.close()
    if found:
        keys = list(found)
        np.savez(f"{out}/{i}.npz", **{f"d{j}": found[k] for j, k in enumerate(keys)})
        index[str(i)] = {"file": fn, "keys": keys}
json.dump(index, open(f"{out}/index.json", "w"))
''')
    r = subprocess.run([CONDA_PY, str(dump), str(tmp_path)], capture_output=True, text=True, timeout=300)
    if r.returncode != 0 or not (tmp_path / "index.json").exists():
        pytest.skip("h5py is not usable here: " + r.stderr[-300:])
    index = json.load(open(tmp_path / "index.json"))
    ok, refused = 0, []
    for i, v in index.items():
        z = np.load(tmp_path / f"{i}.npz")
        for j, key in enumerate(v["keys"]):
            want = z[f"d{j}"]
            try:
                got = jf.hdf5_read(v["file"], key)
            except jf.JfError as ex:
                refused.append((os.path.basename(v["file"]), key, str(ex)))
                continue
            assert got.shape == want.shape and np.array_equal(got, want, equal_nan=True), (v["file"], key)
            ok += 1
    if not index:
        pytest.skip("no third-party HDF5 files in this image")
    assert ok >= 50 and len(refused) <= 2, refused


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="the image's interpreter with h5py is not there")
def test_files_h5py_writes_on_the_spot(tmp_path):
    """What the committed containers do not reach: the format bounds of HDF5 1.8 (superblock 2 with version-3 layouts), a group
    of 300 links and 200 attributes (version-2 B-trees of depth 1 over fractal heaps with several rows of direct blocks), and
    the one refusal a reader of SOFA files should know about: an unlimited dimension written with the 1.10 format."""
    script = tmp_path / "write.py"
    script.write_text('''
import sys, warnings
warnings.simplefilter("ignore")
import h5py, numpy as np
d = sys.argv[1]
rng = np.random.default_rng(1)
a = rng.standard_normal((50, 2, 40))
b = rng.integers(-1000, 1000, (300, 7)).astype(np.int16)
np.savez(d + "/expected.npz", a=a, b=b)
for tag, lv in (("v108", "v108"), ("e_v108", ("earliest", "v108")), ("v110", ("v110", "v110"))):
    with h5py.File(d + "/t_" + tag + ".h5", "w", libver=lv, track_order=True) as f:
        for i in range(12):
            f.attrs.create("attr%d" % i, np.bytes_(b"value%d\\0" % i))
        f.create_dataset("a", data=a, chunks=(7, 2, 16), compression="gzip", shuffle=True, fletcher32=True)
        f.create_dataset("b", data=b, chunks=(64, 7), maxshape=(None, 7))
        f.create_dataset("c", data=b[:5])
        f.create_group("g").create_dataset("x", data=np.arange(10.))
        for i in range(20):
            f.create_dataset("v%d" % i, data=np.full(3, i, np.float32))
for lv, tr in (("earliest", True), ("latest", False)):
    with h5py.File(d + "/many_" + lv + ".h5", "w", libver=lv, track_order=tr) as f:
        for i in range(300):
            f.create_dataset("variable_with_a_long_name_%03d" % i, data=np.full(2, i, np.int32))
        for i in range(200):
            f.attrs.create("attribute_%03d" % i, np.bytes_(b"value %03d" % i))
''')
    r = subprocess.run([CONDA_PY, str(script), str(tmp_path)], capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        pytest.skip("h5py is not usable here: " + r.stderr[-300:])
    E = np.load(tmp_path / "expected.npz")
    for tag in ("v108", "e_v108", "v110"):
        fn = str(tmp_path / f"t_{tag}.h5")
        assert np.array_equal(jf.hdf5_read(fn, "a"), E["a"])
        if tag == "v110":
            with pytest.raises(jf.JfError) as ex:
                jf.hdf5_read(fn, "b")
            assert ex.value.code == jf.JF_ERR_IO and "unlimited" in str(ex.value)
        else:
            assert np.array_equal(jf.hdf5_read(fn, "b"), E["b"].astype(np.float64))
        assert np.array_equal(jf.hdf5_read(fn, "c"), E["b"][:5].astype(np.float64))
        assert jf.hdf5_read(fn, "g/x").tolist() == list(range(10)) and jf.hdf5_read(fn, "v19").tolist() == [19.0] * 3
        assert jf.hdf5_attr(fn, "/", "attr11") == "value11"
    for lv in ("earliest", "latest"):
        fn = str(tmp_path / f"many_{lv}.h5")
        assert all(jf.hdf5_read(fn, "variable_with_a_long_name_%03d" % i).tolist() == [i, i] for i in range(300))
        assert all(jf.hdf5_attr(fn, "/", "attribute_%03d" % i) == "value %03d" % i for i in range(200))
