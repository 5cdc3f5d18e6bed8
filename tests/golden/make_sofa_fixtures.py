"""Writes tests/golden/sofa/*.sofa and sofa_expected.npz: small SOFA-shaped HDF5 files WRITTEN BY libhdf5 (h5py), the pins of
the product's own HDF5 reader (jefferson-2.0_amd/csrc/jf_hdf5.c; tests/test_sofa.py).

Run with an interpreter that has h5py -- in this image /opt/conda/bin/python3.9 (h5py 3.3.0, HDF5 1.10.6); the system
interpreter has none, which is why the files are committed:

    /opt/conda/bin/python3.9 tests/golden/make_sofa_fixtures.py

One HRTF set -- 33 measurements on five rings, 2 receivers, 24 taps, values that are multiples of 2^-15 -- in four containers
that between them use every structure the reader understands:

  nc4.sofa        the way netCDF-4 (the SOFA APIs' library) writes: superblock 0, version-1 object headers, a root group with
                  tracked + indexed link creation order (> 8 links: DENSE link storage -- fractal heap + version-2 B-tree),
                  dimension scales with their reference attributes, 30 fixed-length global string attributes (dense attribute
                  storage), Data.IR float64 chunked + shuffle + deflate (version-1 chunk B-tree, edge chunks), measurements
                  shuffled
  symtab.sofa     the oldest form: symbol-table root group (version-1 group B-tree over several symbol nodes + local heap),
                  float32 contiguous Data.IR, variable-length string attributes (global heap), measurements in ring order
  latest.sofa     libver=latest: superblock 3, version-2 object headers, compact links / dense links, layout version 4: big-endian
                  float32 Data.IR chunked + deflate + fletcher32 (FIXED ARRAY index, filtered), SourcePosition chunked without
                  filters (fixed array), Data.Delay a single filtered chunk, Data.SamplingRate compact, and `paged`: 1100 chunks
                  of one int16 (a PAGED fixed array)
  mono.sofa       four measurements, ONE receiver (refused by jf_sofa_table)
  cartesian.sofa  a 512-byte user block in front of the superblock, Cartesian SourcePosition, big-endian float64 Data.IR,
                  per-measurement integer Data.Delay
"""
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "sofa")

RING_ELE = [-30.0, 0.0, 30.0, 60.0, 90.0]
RING_COUNT = [8, 12, 8, 4, 1]
N = 24


def the_set():
    """(ir [M][2][N] float32, azimuth_sofa [M], elevation [M]) in ring order, azimuth ascending in the TABLE's sense
    (clockwise, 90 = right: 360 - the SOFA azimuth)"""
    rng = np.random.default_rng(2025)
    el, az_table = [], []
    for e, n in zip(RING_ELE, RING_COUNT):
        for i in range(n):
            el.append(e)
            az_table.append(i * 360.0 / n)
    M = len(el)
    ir = rng.standard_normal((M, 2, N)) * np.exp(-np.arange(N) / 5.0)
    ir = np.round(ir * 0.3 * 32768) / 32768          # multiples of 2^-15: exact in float32 and float64
    az_sofa = (360.0 - np.array(az_table)) % 360.0
    return ir.astype(np.float32), az_sofa, np.array(el)


GLOBAL_ATTRS = [
    ("Conventions", "SOFA"), ("Version", "1.0"), ("SOFAConventions", "SimpleFreeFieldHRIR"), ("SOFAConventionsVersion", "1.0"),
    ("APIName", "make_sofa_fixtures.py"), ("APIVersion", "1"), ("ApplicationName", "jefferson tests"), ("ApplicationVersion", "5"),
    ("AuthorContact", "nobody"), ("Comment", "synthetic set on five rings"), ("DataType", "FIR"), ("History", ""),
    ("License", "none"), ("Organization", "none"), ("References", ""), ("RoomType", "free field"), ("Origin", "synthetic"),
    ("DateCreated", "2026-10-05 00:00:00"), ("DateModified", "2026-10-05 00:00:00"), ("Title", "ring33"),
    ("DatabaseName", "ring33"), ("ListenerShortName", "nobody"), ("ListenerDescription", ""), ("ReceiverDescription", ""),
    ("SourceDescription", ""), ("EmitterDescription", ""), ("RoomDescription", ""), ("Extra1", "x"), ("Extra2", "y"),
    ("Extra3", "z"),
]


def fixed(s):
    return np.bytes_(s.encode() + b"\0")          # netCDF-4 writes NC_CHAR attributes as fixed-length, null-terminated


def common_variables(f, M, kw=None, str_attr=fixed):
    kw = kw or {}
    f.create_dataset("ListenerPosition", data=np.zeros((1, 3)), **kw)
    f["ListenerPosition"].attrs["Type"] = str_attr("cartesian")
    f["ListenerPosition"].attrs["Units"] = str_attr("metre")
    f.create_dataset("ListenerUp", data=np.array([[0.0, 0.0, 1.0]]), **kw)
    f.create_dataset("ListenerView", data=np.array([[1.0, 0.0, 0.0]]), **kw)
    f.create_dataset("ReceiverPosition", data=np.array([[[0.0], [0.09], [0.0]], [[0.0], [-0.09], [0.0]]]), **kw)
    f.create_dataset("EmitterPosition", data=np.zeros((1, 3, 1)), **kw)


def write_nc4(path, ir, az, el, order):
    with h5py.File(path, "w", libver="earliest", track_order=True) as f:
        for k, v in GLOBAL_ATTRS:
            f.attrs.create(k, fixed(v))
        M = len(order)
        dims = {"I": 1, "C": 3, "R": 2, "E": 1, "N": N, "M": M, "S": 0}
        for name, n in dims.items():
            d = f.create_dataset(name, data=np.zeros(n, np.float32), track_order=True)
            d.make_scale(name)
        kw = dict(track_order=True)
        d = f.create_dataset("Data.IR", data=ir[order].astype(np.float64), chunks=(16, 1, N), compression="gzip",
                             compression_opts=4, shuffle=True, **kw)
        for i, name in enumerate("MRN"):
            d.dims[i].attach_scale(f[name])
        d = f.create_dataset("Data.SamplingRate", data=np.array([44100.0]), **kw)
        d.attrs.create("Units", fixed("hertz"))
        d.dims[0].attach_scale(f["I"])
        d = f.create_dataset("Data.Delay", data=np.zeros((1, 2)), **kw)
        pos = np.stack([az[order], el[order], np.full(M, 1.4)], axis=1)
        d = f.create_dataset("SourcePosition", data=pos, **kw)
        d.attrs.create("Type", fixed("spherical"))
        d.attrs.create("Units", fixed("degree, degree, metre"))
        d.dims[0].attach_scale(f["M"])
        d.dims[1].attach_scale(f["C"])
        common_variables(f, M, kw)


def write_symtab(path, ir, az, el):
    with h5py.File(path, "w", libver="earliest") as f:
        for k, v in GLOBAL_ATTRS[:12]:
            f.attrs[k] = v                      # variable-length strings: the global heap
        M = len(az)
        f.create_dataset("Data.IR", data=ir)    # float32, contiguous
        f.create_dataset("Data.SamplingRate", data=np.float64(44100.0))     # a scalar dataspace
        f.create_dataset("Data.Delay", data=np.zeros((1, 2), np.float32))
        d = f.create_dataset("SourcePosition", data=np.stack([az, el, np.full(M, 1.4)], axis=1))
        d.attrs["Type"] = "spherical"
        d.attrs["Units"] = "degree, degree, metre"
        common_variables(f, M, str_attr=str)
        for name in ("M", "R", "N", "E", "I", "C", "S", "SourceUp", "SourceView", "RoomCorner", "Aux1", "Aux2", "Aux3"):
            f.create_dataset(name, data=np.zeros(1, np.float32))
        g = f.create_group("nested")           # (a path of two links for the lookup)
        g.create_dataset("seven", data=np.arange(7, dtype=np.int32) - 3)
        g.create_dataset("bytes", data=np.array([200, 3], np.uint8))
        g.create_dataset("wide", data=np.array([-2 ** 40, 2 ** 40], np.int64))


def write_latest(path, ir, az, el, order):
    with h5py.File(path, "w", libver="latest") as f:
        for k, v in GLOBAL_ATTRS:
            f.attrs.create(k, fixed(v))
        M = len(order)
        f.create_dataset("Data.IR", data=ir[order].astype(">f4"), chunks=(8, 2, N), compression="gzip", fletcher32=True)
        pos = np.stack([az[order], el[order], np.full(M, 1.4)], axis=1)
        d = f.create_dataset("SourcePosition", data=pos, chunks=(4, 3))
        d.attrs.create("Type", fixed("spherical"))
        d.attrs.create("Units", fixed("degree, degree, metre"))
        f.create_dataset("Data.Delay", data=np.zeros((1, 2)), chunks=(1, 2), compression="gzip")
        # a compact dataset through the low-level API
        space = h5py.h5s.create_simple((1,))
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_layout(h5py.h5d.COMPACT)
        dsid = h5py.h5d.create(f.id, b"Data.SamplingRate", h5py.h5t.IEEE_F64LE, space, dcpl)
        dsid.write(h5py.h5s.ALL, h5py.h5s.ALL, np.array([44100.0]))
        f.create_dataset("paged", data=(np.arange(1100) * 7 % 1001 - 500).astype(np.int16), chunks=(1,))
        common_variables(f, M)


def write_cartesian(path, ir, az, el, order, delay):
    with h5py.File(path, "w", libver="earliest", userblock_size=512) as f:
        for k, v in GLOBAL_ATTRS[:12]:
            f.attrs.create(k, fixed(v))
        M = len(order)
        f.create_dataset("Data.IR", data=ir[order].astype(">f8"))
        r = 2.0
        a, e = np.radians(az[order]), np.radians(el[order])
        xyz = np.stack([r * np.cos(e) * np.cos(a), r * np.cos(e) * np.sin(a), r * np.sin(e)], axis=1)
        d = f.create_dataset("SourcePosition", data=xyz)
        d.attrs.create("Type", fixed("cartesian"))
        d.attrs.create("Units", fixed("metre"))
        f.create_dataset("Data.SamplingRate", data=np.array([44100], np.int32))
        f.create_dataset("Data.Delay", data=delay[order].astype(np.float64))
        common_variables(f, M)
    with open(path, "r+b") as fp:
        fp.write(b"user block of 512 bytes")


def write_mono(path):
    """one receiver: a set the engine cannot render"""
    with h5py.File(path, "w", libver="earliest") as f:
        f.attrs.create("DataType", fixed("FIR"))
        f.create_dataset("Data.IR", data=np.ones((4, 1, 4), np.float32))
        f.create_dataset("SourcePosition", data=np.array([[a, 0.0, 1.0] for a in (0.0, 90.0, 180.0, 270.0)]))
        f.create_dataset("Data.SamplingRate", data=np.array([44100.0]))


def main():
    os.makedirs(OUT, exist_ok=True)
    ir, az, el = the_set()
    M = len(az)
    rng = np.random.default_rng(7)
    order_a, order_c, order_d = rng.permutation(M), rng.permutation(M), rng.permutation(M)
    delay = rng.integers(0, 5, size=(M, 2))
    write_nc4(os.path.join(OUT, "nc4.sofa"), ir, az, el, order_a)
    write_symtab(os.path.join(OUT, "symtab.sofa"), ir, az, el)
    write_latest(os.path.join(OUT, "latest.sofa"), ir, az, el, order_c)
    write_cartesian(os.path.join(OUT, "cartesian.sofa"), ir, az, el, order_d, delay)
    write_mono(os.path.join(OUT, "mono.sofa"))
    np.savez(os.path.join(OUT, "sofa_expected.npz"), ir=ir, az_sofa=az.astype(np.float64), el=el, order_nc4=order_a,
             order_latest=order_c, order_cartesian=order_d, delay=delay, ring_ele=np.array(RING_ELE), ring_count=np.array(RING_COUNT),
             paged=(np.arange(1100) * 7 % 1001 - 500).astype(np.int16))
    for n in sorted(os.listdir(OUT)):
        print(n, os.path.getsize(os.path.join(OUT, n)))
    print("h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version, "python", sys.version.split()[0])


if __name__ == "__main__":
    main()
