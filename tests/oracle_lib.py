"""ctypes binding of oracle/libjf_oracle.so -- the CPU checker (tests/bench baseline only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libjf_oracle.so")

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int)


def build(force=False):
    src = [os.path.join(ORACLE_DIR, n) for n in ("jf_oracle.c", "jf_oracle.h", "Makefile")]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return LIB_PATH
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "libjf_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.jfo_azimuth_offsets.argtypes = [_i]
        L.jfo_table_positions.argtypes = [_i, _i]
        L.jfo_pick_hrtf.argtypes = [C.c_float, C.c_float]
        L.jfo_pick_hrtf.restype = C.c_int
        L.jfo_interp.argtypes = [C.c_float, C.c_float, _i, _f]
        L.jfo_interp.restype = C.c_int
        L.jfo_interp_corrected.argtypes = [C.c_float, C.c_float, _i, _f]
        L.jfo_interp_corrected.restype = C.c_int
        L.jfo_case.argtypes = [_i]
        L.jfo_case.restype = C.c_int
        L.jfo_terms.argtypes = [_i, _f, _i, _f]
        L.jfo_terms.restype = C.c_int
        L.jfo_from_spherical.argtypes = [C.c_float] * 3 + [_f]
        L.jfo_from_cartesian.argtypes = [C.c_float] * 3 + [_f]
        L.jfo_from_cartesian.restype = C.c_int
        L.jfo_distance_factor.argtypes = [C.c_float] * 3 + [C.c_int, _f]
        L.jfo_build_table.argtypes = [_f, C.c_int, C.c_int, C.c_int, _f]
        L.jfo_rfft.argtypes = [_f, C.c_int, _f]
        L.jfo_irfft.argtypes = [_f, C.c_int, _f]
        L.jfo_create.argtypes = [C.c_int, C.c_int, C.c_int, _f, C.c_int]
        L.jfo_create.restype = C.c_void_p
        L.jfo_destroy.argtypes = [C.c_void_p]
        L.jfo_pad_len.argtypes = [C.c_void_p]
        L.jfo_pad_len.restype = C.c_int
        L.jfo_set_mode.argtypes = [C.c_void_p, C.c_int]
        L.jfo_source_set_signal.argtypes = [C.c_void_p, C.c_int, _f, C.c_int]
        L.jfo_source_set_signal.restype = C.c_int
        L.jfo_source_set_spherical.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 3
        L.jfo_source_set_spherical.restype = C.c_int
        L.jfo_source_set_cartesian.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 3
        L.jfo_source_set_cartesian.restype = C.c_int
        L.jfo_source_reset.argtypes = [C.c_void_p, C.c_int]
        L.jfo_process_block.argtypes = [C.c_void_p, _f]
        L.jfo_source_last_block.argtypes = [C.c_void_p, C.c_int]
        L.jfo_source_last_block.restype = _f
        L.jfo_process_batch.argtypes = [C.c_void_p, C.c_int, _f, _f, _f, C.c_int]
        L.jfo_num_threads.restype = C.c_int
        L.jfo_reverb_set_ir.argtypes = [C.c_void_p, _f, C.c_int, C.c_float]
        L.jfo_reverb_set_ir.restype = C.c_int
        L.jfo_reverb_padded_size.argtypes = [C.c_int, C.c_int]
        L.jfo_reverb_padded_size.restype = C.c_int
        L.jfo_reverb_offline.argtypes = [_f, C.c_int, _f, C.c_int, _f]
        L.jfo_reverb_offline.restype = C.c_float
        L.jfo_grid_rows.argtypes = [C.c_int, _i]
        L.jfo_grid_rows.restype = C.c_int
        L.jfo_grid_interp.argtypes = [C.c_int, _f, _i, _f, C.c_float, C.c_float, _i, _f]
        L.jfo_grid_interp.restype = C.c_int
        L.jfo_grid_pick.argtypes = [C.c_int, _f, _i, _f, C.c_float, C.c_float]
        L.jfo_grid_pick.restype = C.c_int
        L.jfo_kemar_grid.argtypes = [_f, _i, _f]
        L.jfo_create_grid.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _f, _i, _f, _f, C.c_int]
        L.jfo_create_grid.restype = C.c_void_p
        _lib = L
    return _lib


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f)


def iptr(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_i)


def azimuth_offsets():
    a = np.zeros(15, np.int32)
    lib().jfo_azimuth_offsets(iptr(a))
    return a.tolist()


def table_positions():
    e = np.zeros(710, np.int32)
    a = np.zeros(710, np.int32)
    lib().jfo_table_positions(iptr(e), iptr(a))
    return list(zip(e.tolist(), a.tolist()))


def pick_hrtf(ele, azi):
    return lib().jfo_pick_hrtf(ele, azi)


def interp(ele, azi, corrected=False):
    idx = np.zeros(4, np.int32)
    om = np.zeros(6, np.float32)
    f = lib().jfo_interp_corrected if corrected else lib().jfo_interp
    if f(ele, azi, iptr(idx), fptr(om)):
        return None
    return idx, om


def terms(idx, om):
    rows = np.zeros(4, np.int32)
    w = np.zeros(4, np.float32)
    n = lib().jfo_terms(iptr(np.ascontiguousarray(idx, np.int32)),
                        fptr(np.ascontiguousarray(om, np.float32)), iptr(rows), fptr(w))
    return rows[:n].copy(), w[:n].copy()


def from_spherical(ele, azi, r):
    o = np.zeros(5, np.float32)
    lib().jfo_from_spherical(ele, azi, r, fptr(o))
    return o


def from_cartesian(x, y, z):
    o = np.zeros(3, np.float32)
    if lib().jfo_from_cartesian(x, y, z, fptr(o)):
        return None
    return o


def distance_factor(x, y, z, nc):
    d = np.zeros(2 * nc, np.float32)
    lib().jfo_distance_factor(x, y, z, nc, fptr(d))
    return d.view(np.complex64)


def build_table(hrir, N):
    hrir = np.ascontiguousarray(hrir, np.float32)
    n, _, taps = hrir.shape
    t = np.zeros((n, 2, N // 2 + 1, 2), np.float32)
    lib().jfo_build_table(fptr(hrir), n, taps, N, fptr(t))
    return t.view(np.complex64)[..., 0]


def rfft(x):
    x = np.ascontiguousarray(x, np.float32)
    X = np.zeros((len(x) // 2 + 1) * 2, np.float32)
    lib().jfo_rfft(fptr(x), len(x), fptr(X))
    return X.view(np.complex64)


def irfft(X, N):
    X = np.ascontiguousarray(X, np.complex64)
    y = np.zeros(N, np.float32)
    lib().jfo_irfft(fptr(X.view(np.float32)), N, fptr(y))
    return y


def reverb_offline(x, ir):
    """cudaPart.cu:87-172 as jfo_reverb_offline restates it: (buf of new_size samples, rms gain)."""
    x = np.ascontiguousarray(x, np.float32)
    ir = np.ascontiguousarray(ir, np.float32)
    out = np.zeros(lib().jfo_reverb_padded_size(len(x), len(ir)), np.float32)
    g = lib().jfo_reverb_offline(fptr(x), len(x), fptr(ir), len(ir), fptr(out))
    return out, float(g)


class Grid:
    """(ring elevations, counts, steps or None) as the C oracle takes them"""

    def __init__(self, ring_ele, ring_count, ring_step=None):
        self.ele = np.ascontiguousarray(ring_ele, np.float32)
        self.count = np.ascontiguousarray(ring_count, np.int32)
        self.step = None if ring_step is None else np.ascontiguousarray(ring_step, np.float32)
        self.n = len(self.ele)
        self.n_rows = int(self.count.sum())

    def args(self):
        return self.n, fptr(self.ele), iptr(self.count), fptr(self.step) if self.step is not None else None

    @staticmethod
    def kemar():
        e, c, s = np.zeros(14, np.float32), np.zeros(14, np.int32), np.zeros(14, np.float32)
        lib().jfo_kemar_grid(fptr(e), iptr(c), fptr(s))
        return Grid(e, c, s)

    def interp(self, ele, azi):
        idx = np.zeros(4, np.int32)
        om = np.zeros(6, np.float32)
        rc = lib().jfo_grid_interp(*self.args(), ele, azi, iptr(idx), fptr(om))
        assert rc != -2, "bad grid"
        return None if rc else (idx, om)

    def pick(self, ele, azi):
        return lib().jfo_grid_pick(*self.args(), ele, azi)


class Engine:
    """Same method names as the HIP engine binding so tests read alike."""

    def __init__(self, B, hrtf_len, n_sources, hrir, grid=None):
        self.hrir = np.ascontiguousarray(hrir, np.float32)
        self.B, self.S = B, n_sources
        if grid is None:
            self.h = lib().jfo_create(B, hrtf_len, n_sources, fptr(self.hrir), self.hrir.shape[2])
        else:
            assert self.hrir.shape[0] == grid.n_rows
            self.h = lib().jfo_create_grid(B, hrtf_len, n_sources, *grid.args(), fptr(self.hrir), self.hrir.shape[2])
        if not self.h:
            raise ValueError("jfo_create failed")
        self.N = lib().jfo_pad_len(self.h)

    def close(self):
        if self.h:
            lib().jfo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_mode(self, mode):
        lib().jfo_set_mode(self.h, int(mode))

    def set_signal(self, s, mono):
        mono = np.ascontiguousarray(mono, np.float32)
        assert lib().jfo_source_set_signal(self.h, s, fptr(mono), len(mono)) == 0

    def set_spherical(self, s, ele, azi, r):
        assert lib().jfo_source_set_spherical(self.h, s, ele, azi, r) == 0

    def set_cartesian(self, s, x, y, z):
        return lib().jfo_source_set_cartesian(self.h, s, x, y, z)

    def reset(self, s):
        lib().jfo_source_reset(self.h, s)

    def set_reverb(self, ir, gain=1.0):
        ir = np.ascontiguousarray(ir, np.float32)
        assert lib().jfo_reverb_set_ir(self.h, fptr(ir) if len(ir) else None, len(ir), gain) == 0

    def process_block(self):
        out = np.zeros(2 * self.B, np.float32)
        lib().jfo_process_block(self.h, fptr(out))
        return out

    def last_block(self, s):
        p = lib().jfo_source_last_block(self.h, s)
        return np.ctypeslib.as_array(p, shape=(2 * self.B,)).copy()

    def process_batch(self, pos, want_partial=False, n_threads=0):
        pos = np.ascontiguousarray(pos, np.float32)
        K, S = pos.shape[0], pos.shape[1]
        assert S == self.S and pos.shape[2] == 5
        mix = np.zeros((K, 2 * self.B), np.float32)
        part = np.zeros((S, K, 2 * self.B), np.float32) if want_partial else None
        lib().jfo_process_batch(self.h, K, fptr(pos), fptr(mix),
                                fptr(part) if want_partial else None, n_threads)
        return (mix, part) if want_partial else mix
