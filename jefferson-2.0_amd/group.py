"""ctypes binding of libjefferson_group.so (include/jefferson_group.h): test plumbing, like __init__.py.
The product is the C library; a C host calls it directly."""
import ctypes as C
import os

import numpy as np

from . import JfConfig, JfError, JfHrtfGrid, _f, _fp, lib as core_lib, NUM_HRTF

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjefferson_group.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "jefferson_group.h")

_SIGS = {
    "jf_shard_range": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "jf_group_create": (C.c_int, [C.POINTER(JfConfig), C.c_int, C.POINTER(C.c_int), _f, C.c_int, C.POINTER(C.c_void_p)]),
    "jf_group_create_grid": (C.c_int, [C.POINTER(JfConfig), C.c_int, C.POINTER(C.c_int), C.POINTER(JfHrtfGrid), _f, C.c_int,
                                       C.POINTER(C.c_void_p)]),
    "jf_group_create_sofa": (C.c_int, [C.POINTER(JfConfig), C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_float, C.POINTER(C.c_void_p)]),
    "jf_group_destroy": (None, [C.c_void_p]),
    "jf_group_last_error": (C.c_char_p, [C.c_void_p]),
    "jf_group_num_gpus": (C.c_int, [C.c_void_p]),
    "jf_group_num_sources": (C.c_int, [C.c_void_p]),
    "jf_group_engine": (C.c_void_p, [C.c_void_p, C.c_int]),
    "jf_group_first_source": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_group_source_set_signal": (C.c_int, [C.c_void_p, C.c_int, _f, C.c_size_t]),
    "jf_group_source_set_spherical": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float]),
    "jf_group_source_set_cartesian": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float]),
    "jf_group_process_block": (C.c_int, [C.c_void_p, _f]),
    "jf_group_process_batch": (C.c_int, [C.c_void_p, C.c_int, _f, _f]),
    "jf_group_batch_upload_positions": (C.c_int, [C.c_void_p, C.c_int, _f]),
    "jf_group_batch_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "jf_group_batch_fetch": (C.c_int, [C.c_void_p, _f]),
    "jf_group_synchronize": (C.c_int, [C.c_void_p]),
    "jf_group_source_reset": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_group_set_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_group_set_pause": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_group_reverb_set_ir": (C.c_int, [C.c_void_p, _f, C.c_size_t, C.c_float]),
    "jf_group_last_block_peak": (C.c_float, [C.c_void_p]),
    "jf_group_failed": (C.c_int, [C.c_void_p]),
    "jf_group_create_shards_on_device": (C.c_int, [C.POINTER(JfConfig), C.c_int, C.c_int, _f, C.c_int, C.POINTER(C.c_void_p)]),
    "jf_group_debug_fail_next": (C.c_int, [C.c_void_p, C.c_int]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JfError(-3, f"{LIB_PATH} is not built (run __graft_entry__.build())")
        core_lib()  # libjefferson_hip.so first, from the same directory
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def shard_range(n_total, n_parts, part):
    lo, hi = C.c_int(), C.c_int()
    rc = lib().jf_shard_range(n_total, n_parts, part, C.byref(lo), C.byref(hi))
    return None if rc else (lo.value, hi.value)


class Group:
    def __init__(self, B, hrtf_len, n_sources, hrir, n_gpus=1, devices=None, max_batch_blocks=1, flags=0, shards_on_device=0,
                 grid=None, sofa=None, sofa_tol_deg=0.05):
        """shards_on_device = n > 0: n shards of the job on device 0 with a host sum instead of RCCL (test support,
        jf_group_create_shards_on_device).  grid: a jf.Grid of the HRTF set's own (jf_group_create_grid).  sofa: the set of a
        SOFA file instead of hrir (jf_group_create_sofa)."""
        L = lib()
        cfg = JfConfig(B, hrtf_len, n_sources, 0, max_batch_blocks, flags)
        h = C.c_void_p()
        dev = (C.c_int * n_gpus)(*devices) if devices is not None else None
        if sofa is None:
            hrir = np.ascontiguousarray(hrir, np.float32)
            assert hrir.ndim == 3 and hrir.shape[0] == (grid.rows() if grid is not None else NUM_HRTF) and hrir.shape[1] == 2
        if sofa is not None:
            rc = L.jf_group_create_sofa(C.byref(cfg), n_gpus, dev, os.fsencode(sofa), sofa_tol_deg, C.byref(h))
        elif grid is not None:
            self._grid = grid
            rc = L.jf_group_create_grid(C.byref(cfg), n_gpus, dev, C.byref(grid.c), _fp(hrir), hrir.shape[2], C.byref(h))
        elif shards_on_device > 0:
            rc = L.jf_group_create_shards_on_device(C.byref(cfg), int(shards_on_device), 0, _fp(hrir), hrir.shape[2], C.byref(h))
        else:
            rc = L.jf_group_create(C.byref(cfg), n_gpus, dev, _fp(hrir), hrir.shape[2], C.byref(h))
        if rc:
            raise JfError(rc, L.jf_group_last_error(None).decode())
        self.h, self.B, self.S, self.maxK = h, B, n_sources, max_batch_blocks

    def close(self):
        if getattr(self, "h", None):
            lib().jf_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise JfError(rc, lib().jf_group_last_error(self.h).decode())

    def num_gpus(self):
        return lib().jf_group_num_gpus(self.h)

    def first_source(self, i):
        return lib().jf_group_first_source(self.h, i)

    def set_signal(self, s, mono):
        mono = np.ascontiguousarray(mono, np.float32)
        self._chk(lib().jf_group_source_set_signal(self.h, s, _fp(mono), len(mono)))

    def set_spherical(self, s, ele, azi, r):
        return lib().jf_group_source_set_spherical(self.h, s, ele, azi, r)

    def set_cartesian(self, s, x, y, z):
        return lib().jf_group_source_set_cartesian(self.h, s, x, y, z)

    def process_block(self):
        out = np.zeros(2 * self.B, np.float32)
        self._chk(lib().jf_group_process_block(self.h, _fp(out)))
        return out

    def process_batch(self, pos):
        pos = np.ascontiguousarray(pos, np.float32)
        K = pos.shape[0]
        assert pos.shape[1] == self.S and pos.shape[2] == 5
        mix = np.zeros((K, 2 * self.B), np.float32)
        self._chk(lib().jf_group_process_batch(self.h, K, _fp(pos), _fp(mix)))
        return mix

    def upload_positions(self, pos):
        pos = np.ascontiguousarray(pos, np.float32)
        self._chk(lib().jf_group_batch_upload_positions(self.h, pos.shape[0], _fp(pos)))

    def batch_run(self, first, n):
        return lib().jf_group_batch_run(self.h, first, n)

    def batch_fetch(self, n):
        out = np.zeros((n, 2 * self.B), np.float32)
        rc = lib().jf_group_batch_fetch(self.h, _fp(out))
        return rc, out

    def synchronize(self):
        self._chk(lib().jf_group_synchronize(self.h))

    def reset(self, s):
        self._chk(lib().jf_group_source_reset(self.h, s))

    def set_mode(self, mode):
        return lib().jf_group_set_mode(self.h, int(mode))

    def set_pause(self, paused):
        self._chk(lib().jf_group_set_pause(self.h, int(bool(paused))))

    def set_reverb(self, ir, gain=1.0):
        ir = np.ascontiguousarray(ir, np.float32)
        self._chk(lib().jf_group_reverb_set_ir(self.h, _fp(ir) if len(ir) else None, len(ir), gain))

    def last_block_peak(self):
        return float(lib().jf_group_last_block_peak(self.h))

    def failed(self):
        return lib().jf_group_failed(self.h)
