"""jefferson-2.0_amd -- ctypes binding of libjefferson_hip.so (include/jefferson.h).

Plumbing for tests and bench.py; the product is the C ABI.  The directory name is
not an importable identifier, so load it with `jf_load.py` at the repo root
(`from jf_load import jf`).  There is no fallback: if the HIP library is missing
or no GPU is usable, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# JF_LIB selects an A/B build of the same library (csrc/Makefile `variant`); default = the product
LIB_PATH = os.environ.get("JF_LIB") or os.path.join(_HERE, "libjefferson_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "jefferson.h")
DEBUG_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "jefferson_debug.h")  # taps, timing hooks, tuning switches

JF_OK, JF_ERR_ARG, JF_ERR_RANGE, JF_ERR_DEVICE, JF_ERR_IO, JF_ERR_STATE, JF_ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6
JF_FLAG_CORRECTED_INTERPOLATION = 1
JF_FLAG_NO_INTERP_TABLE = 2
INTERP_ROWS = 131 * 360
JF_MODE_FD_COMPLEX, JF_MODE_FD_BASIC = 0, 1
NUM_HRTF = 710
PAD_LEN = 1024
NC = 513

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int)


class JfConfig(C.Structure):
    _fields_ = [("frames_per_buffer", C.c_int), ("hrtf_len", C.c_int), ("n_sources", C.c_int),
                ("device", C.c_int), ("max_batch_blocks", C.c_int), ("flags", C.c_uint)]


class JfHrtfGrid(C.Structure):
    _fields_ = [("n_rings", C.c_int), ("ring_elevation", C.POINTER(C.c_float)), ("ring_count", C.POINTER(C.c_int)),
                ("ring_step", C.POINTER(C.c_float))]


JF_MAX_RINGS = 40


class JfGridLayout(C.Structure):
    _fields_ = [("n_rings", C.c_int), ("ring_elevation", C.c_float * JF_MAX_RINGS), ("ring_count", C.c_int * JF_MAX_RINGS),
                ("ring_step", C.c_float * JF_MAX_RINGS)]


class JfSofaSet(C.Structure):
    _fields_ = [("n_measurements", C.c_int), ("n_receivers", C.c_int), ("n_samples", C.c_int), ("sample_rate", C.c_double),
                ("ir", C.POINTER(C.c_float)), ("azimuth", C.POINTER(C.c_float)), ("elevation", C.POINTER(C.c_float)),
                ("distance", C.POINTER(C.c_float)), ("delay", C.POINTER(C.c_float)), ("conventions", C.c_char * 48)]


class JfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"jefferson error {code}: {msg}")
        self.code = code


_lib = None

_SIGS = {
    "jf_engine_create": (C.c_int, [C.POINTER(JfConfig), _f, C.c_int, C.POINTER(C.c_void_p)]),
    "jf_engine_create_from_dir": (C.c_int, [C.POINTER(JfConfig), C.c_char_p, C.POINTER(C.c_void_p)]),
    "jf_engine_create_grid": (C.c_int, [C.POINTER(JfConfig), C.POINTER(JfHrtfGrid), _f, C.c_int, C.POINTER(C.c_void_p)]),
    "jf_engine_create_sofa": (C.c_int, [C.POINTER(JfConfig), C.c_char_p, C.c_float, C.POINTER(C.c_void_p)]),
    "jf_sofa_read": (C.c_int, [C.c_char_p, C.POINTER(JfSofaSet)]),
    "jf_sofa_release": (None, [C.POINTER(JfSofaSet)]),
    "jf_sofa_taps": (C.c_int, [C.POINTER(JfSofaSet)]),
    "jf_sofa_table": (C.c_int, [C.POINTER(JfSofaSet), C.c_float, C.c_void_p, _f, C.c_int]),
    "jf_debug_hdf5_read": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.POINTER(C.c_double)), _i, C.POINTER(C.c_ulonglong)]),
    "jf_debug_hdf5_attr": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "jf_kemar_grid": (C.c_int, [C.POINTER(JfHrtfGrid)]),
    "jf_grid_rows": (C.c_int, [C.POINTER(JfHrtfGrid)]),
    "jf_grid_from_positions": (C.c_int, [C.c_size_t, _f, _f, C.c_float, C.c_void_p, _i]),
    "jf_grid_interpolation": (C.c_int, [C.POINTER(JfHrtfGrid), C.c_float, C.c_float, _i, _f]),
    "jf_grid_pick": (C.c_int, [C.POINTER(JfHrtfGrid), C.c_float, C.c_float]),
    "jf_table_rows": (C.c_int, [C.c_void_p]),
    "jf_engine_destroy": (None, [C.c_void_p]),
    "jf_last_error": (C.c_char_p, [C.c_void_p]),
    "jf_frames_per_buffer": (C.c_int, [C.c_void_p]),
    "jf_pad_len": (C.c_int, [C.c_void_p]),
    "jf_num_sources": (C.c_int, [C.c_void_p]),
    "jf_source_set_signal": (C.c_int, [C.c_void_p, C.c_int, _f, C.c_size_t]),
    "jf_source_set_cartesian": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float]),
    "jf_source_set_spherical": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float]),
    "jf_source_get_position": (C.c_int, [C.c_void_p, C.c_int, _f]),
    "jf_source_reset": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_position_from_spherical": (C.c_int, [C.c_float, C.c_float, C.c_float, _f]),
    "jf_position_from_cartesian": (C.c_int, [C.c_float, C.c_float, C.c_float, _f]),
    "jf_positions_from_spherical": (C.c_int, [C.c_size_t, _f, _f, _f, _f]),
    "jf_interpolation": (C.c_int, [C.c_float, C.c_float, _i, _f]),
    "jf_interpolation_ex": (C.c_int, [C.c_float, C.c_float, C.c_uint, _i, _f]),
    "jf_pick_hrtf": (C.c_int, [C.c_float, C.c_float]),
    "jf_process_block": (C.c_int, [C.c_void_p, _f]),
    "jf_submit_block": (C.c_int, [C.c_void_p]),
    "jf_collect_block": (C.c_int, [C.c_void_p, _f]),
    "jf_callback": (C.c_int, [C.c_void_p, _f]),
    "jf_pa_callback": (C.c_int, [C.c_void_p, C.c_void_p, C.c_ulong, C.c_void_p, C.c_ulong, C.c_void_p]),
    "jf_set_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_set_pause": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_process_batch": (C.c_int, [C.c_void_p, C.c_int, _f, _f]),
    "jf_batch_upload_positions": (C.c_int, [C.c_void_p, C.c_int, _f]),
    "jf_batch_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "jf_synchronize": (C.c_int, [C.c_void_p]),
    "jf_batch_fetch": (C.c_int, [C.c_void_p, C.c_int, _f]),
    "jf_batch_mix_device": (C.c_void_p, [C.c_void_p]),
    "jf_batch_partial_device": (C.c_void_p, [C.c_void_p]),
    "jf_engine_stream": (C.c_void_p, [C.c_void_p]),
    "jf_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "jf_reverb_set_ir": (C.c_int, [C.c_void_p, _f, C.c_size_t, C.c_float]),
    "jf_reverb_rms_gain": (C.c_float, [_f, C.c_size_t, _f, C.c_size_t]),
    "jf_profile_read_reverb": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "jf_profile_set_stride": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_debug_set_source_group": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_debug_read_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]),
    "jf_debug_source_order": (C.c_int, [C.c_void_p, _i]),
    "jf_debug_set_reverb_form": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_last_block_peak": (C.c_float, [C.c_void_p]),
    "jf_debug_last_kernels": (C.c_char_p, [C.c_void_p]),
    "jf_debug_last_source_group": (C.c_int, [C.c_void_p]),
    "jf_debug_set_grid_limit": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_debug_set_prep_ahead": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_debug_stage_taps": (C.c_int, [C.c_void_p, C.c_int, _f, _f, _f, _f]),
    "jf_debug_copy_from_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "jf_debug_set_rt_max_sources": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_debug_read_table": (C.c_int, [C.c_void_p, _f]),
    "jf_debug_set_interp_table": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_debug_interp_table": (C.c_int, [C.c_void_p]),
    "jf_debug_interp_table_built": (C.c_int, [C.c_void_p]),
    "jf_debug_set_reverb_partitioning": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_debug_set_reverb_async": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_debug_set_reverb_head_fused": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_debug_set_reverb_lazy_state": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_debug_set_reverb_ahead": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_debug_set_reverb_side_workgroups": (C.c_int, [C.c_void_p, C.c_int]),
    "jf_sources_set_latched": (C.c_int, [C.c_void_p, _f]),
    "jf_device_numa_node": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "jf_pin_thread_to_device": (C.c_int, [C.c_int]),
    "jf_debug_reverb_partitions": (C.c_int, [C.c_void_p, _i, _i, _i]),
    "jf_debug_reverb_schedule": (C.c_int, [C.c_longlong, C.c_int, C.c_int, C.c_longlong, C.POINTER(C.c_longlong)]),
    "jf_debug_last_run_used_rows": (C.c_int, [C.c_void_p]),
    "jf_debug_count_desc_flags": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "jf_debug_read_table_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _f]),
    "jf_debug_interp_device": (C.c_int, [C.c_void_p, C.c_int, _f, _f, _i, _f, _i]),
    "jf_debug_rfft_device": (C.c_int, [C.c_void_p, C.c_int, _f, _f]),
    "jf_wav_read_mono": (C.c_int, [C.c_char_p, C.POINTER(_f), C.POINTER(C.c_size_t), _i]),
    "jf_wav_write_stereo24": (C.c_int, [C.c_char_p, _f, C.c_size_t, C.c_int]),
    "jf_free": (None, [C.c_void_p]),
}


def exported_symbols():
    """Names every entry point of include/jefferson.h and include/jefferson_debug.h must be exported under."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JfError(JF_ERR_DEVICE, f"{LIB_PATH} is not built (run __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f)


def _ip(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_i)


def device_numa_node(device=0):
    """NUMA node of HIP device `device` (-1: the system does not say)."""
    n = C.c_int(-1)
    rc = lib().jf_device_numa_node(int(device), C.byref(n))
    if rc:
        raise JfError(rc, (lib().jf_last_error(None) or b"").decode())
    return n.value


def pin_thread_to_device(device=0):
    """Restrict the calling thread to the CPUs of the device's NUMA node (include/jefferson.h); False if that is not possible."""
    return lib().jf_pin_thread_to_device(int(device)) == 0


def position_from_spherical(ele, azi, r):
    o = np.zeros(5, np.float32)
    rc = lib().jf_position_from_spherical(ele, azi, r, _fp(o))
    if rc:
        raise JfError(rc, "position_from_spherical")
    return o


def positions_from_spherical(ele, azi, r):
    """Vectorised: arrays of equal shape -> float32 [..., 5] latched records."""
    ele = np.ascontiguousarray(ele, np.float32)
    azi = np.ascontiguousarray(np.broadcast_to(azi, ele.shape), np.float32)
    r = np.ascontiguousarray(np.broadcast_to(r, ele.shape), np.float32)
    out = np.zeros(ele.shape + (5,), np.float32)
    rc = lib().jf_positions_from_spherical(ele.size, _fp(ele), _fp(azi), _fp(r), _fp(out))
    if rc:
        raise JfError(rc, "positions_from_spherical")
    return out


def position_from_cartesian(x, y, z):
    o = np.zeros(5, np.float32)
    rc = lib().jf_position_from_cartesian(x, y, z, _fp(o))
    return None if rc else o


def interpolation(ele, azi, flags=0):
    idx = np.zeros(4, np.int32)
    om = np.zeros(6, np.float32)
    rc = lib().jf_interpolation_ex(ele, azi, flags, _ip(idx), _fp(om)) if flags else \
        lib().jf_interpolation(ele, azi, _ip(idx), _fp(om))
    return None if rc else (idx, om)


def reverb_schedule(j0, K, M, fut_m):
    """jf_debug_reverb_schedule as a dict (host logic only)"""
    out = (C.c_longlong * 16)()
    rc = lib().jf_debug_reverb_schedule(j0, K, M, fut_m, out)
    if rc:
        raise JfError(rc, "reverb_schedule")
    names = ("m_lo", "n_tr", "ma", "n_mid", "n_ranges", "kb0", "kn0", "kb1", "kn1", "copy_lo", "copy_hi", "skip_lo", "skip_hi",
             "tail_early", "tail_late", "fut_m")
    return {n: int(v) for n, v in zip(names, out)}


def pick_hrtf(ele, azi):
    return lib().jf_pick_hrtf(ele, azi)


def reverb_rms_gain(signal, ir):
    signal = np.ascontiguousarray(signal, np.float32)
    ir = np.ascontiguousarray(ir, np.float32)
    return float(lib().jf_reverb_rms_gain(_fp(signal), len(signal), _fp(ir), len(ir)))


def wav_read_mono(path):
    p = _f()
    n = C.c_size_t()
    fs = C.c_int()
    rc = lib().jf_wav_read_mono(path.encode(), C.byref(p), C.byref(n), C.byref(fs))
    if rc:
        raise JfError(rc, lib().jf_last_error(None).decode())
    out = np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.float32)
    lib().jf_free(p)
    return out, fs.value


def wav_write_stereo24(path, interleaved, fs=44100):
    a = np.ascontiguousarray(interleaved, np.float32).reshape(-1)
    rc = lib().jf_wav_write_stereo24(path.encode(), _fp(a), len(a) // 2, fs)
    if rc:
        raise JfError(rc, lib().jf_last_error(None).decode())


class SofaSet:
    """include/jefferson.h: jf_sofa_read -- the variables of a SOFA file as arrays (copies; the library's buffers are released)."""

    def __init__(self, path):
        self.c = None
        c = JfSofaSet()
        rc = lib().jf_sofa_read(os.fsencode(path), C.byref(c))
        if rc:
            raise JfError(rc, lib().jf_last_error(None).decode())
        self.c = c
        M, R, N = c.n_measurements, c.n_receivers, c.n_samples
        self.M, self.R, self.N, self.sample_rate = M, R, N, c.sample_rate
        self.conventions = c.conventions.decode(errors="replace")
        arr = lambda p, shape: np.ctypeslib.as_array(p, shape=shape).copy()
        self.ir = arr(c.ir, (M, R, N))
        self.azimuth, self.elevation, self.distance = arr(c.azimuth, (M,)), arr(c.elevation, (M,)), arr(c.distance, (M,))
        self.delay = arr(c.delay, (M, R))

    def taps(self):
        n = lib().jf_sofa_taps(C.byref(self.c))
        if n < 0:
            raise JfError(n, lib().jf_last_error(None).decode())
        return n

    def table(self, tol_deg=0.05, taps=None):
        """(Grid, hrir [M][2][taps]) for Engine(..., hrir=, grid=): include/jefferson.h: jf_sofa_table"""
        taps = self.taps() if taps is None else taps
        lay = JfGridLayout()
        hrir = np.zeros((self.M, 2, taps), np.float32)
        rc = lib().jf_sofa_table(C.byref(self.c), tol_deg, C.byref(lay), _fp(hrir), taps)
        if rc:
            raise JfError(rc, lib().jf_last_error(None).decode())
        n = lay.n_rings
        return Grid(list(lay.ring_elevation[:n]), list(lay.ring_count[:n]), list(lay.ring_step[:n])), hrir

    def close(self):
        if self.c is not None:
            lib().jf_sofa_release(C.byref(self.c))
            self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hdf5_read(path, dataset):
    """tests: a numeric dataset of an HDF5 file through the library's own reader (jf_hdf5.c), float64"""
    p = C.POINTER(C.c_double)()
    rank = C.c_int()
    dims = (C.c_ulonglong * 8)()
    rc = lib().jf_debug_hdf5_read(os.fsencode(path), dataset.encode(), C.byref(p), C.byref(rank), dims)
    if rc:
        raise JfError(rc, lib().jf_last_error(None).decode())
    shape = tuple(int(dims[i]) for i in range(rank.value))
    n = int(np.prod(shape)) if shape else 1
    out = np.ctypeslib.as_array(p, shape=(n,)).copy().reshape(shape) if n else np.zeros(shape)
    lib().jf_free(p)
    return out


def hdf5_attr(path, obj, attr):
    """tests: a string attribute (None if the object has none of that name)"""
    buf = C.create_string_buffer(512)
    rc = lib().jf_debug_hdf5_attr(os.fsencode(path), obj.encode(), attr.encode(), buf, 512)
    if rc == JF_ERR_ARG:
        return None
    if rc:
        raise JfError(rc, lib().jf_last_error(None).decode())
    return buf.value.decode(errors="replace")


class Grid:
    """include/jefferson.h: jf_hrtf_grid (keeps its arrays alive)."""

    def __init__(self, ring_elevation, ring_count, ring_step=None):
        self.ele = np.ascontiguousarray(ring_elevation, np.float32)
        self.count = np.ascontiguousarray(ring_count, np.int32)
        self.step = None if ring_step is None else np.ascontiguousarray(ring_step, np.float32)
        assert len(self.ele) == len(self.count) and (self.step is None or len(self.step) == len(self.ele))
        self.c = JfHrtfGrid(len(self.ele), _fp(self.ele), _ip(self.count), _fp(self.step) if self.step is not None else None)

    @staticmethod
    def kemar():
        g = JfHrtfGrid()
        assert lib().jf_kemar_grid(C.byref(g)) == 0
        n = g.n_rings
        return Grid([g.ring_elevation[i] for i in range(n)], [g.ring_count[i] for i in range(n)],
                    [g.ring_step[i] for i in range(n)])

    @staticmethod
    def from_positions(azimuth_deg, elevation_deg, tol_deg=0.05):
        """(Grid, row_of): the rings of a set from its measurements' directions (include/jefferson.h: jf_grid_from_positions)"""
        az = np.ascontiguousarray(azimuth_deg, np.float32)
        el = np.ascontiguousarray(elevation_deg, np.float32)
        assert az.shape == el.shape and az.ndim == 1
        lay = JfGridLayout()
        row_of = np.zeros(len(az), np.int32)
        rc = lib().jf_grid_from_positions(len(az), _fp(az), _fp(el), tol_deg, C.byref(lay), _ip(row_of))
        if rc:
            raise JfError(rc, lib().jf_last_error(None).decode())
        n = lay.n_rings
        return Grid(list(lay.ring_elevation[:n]), list(lay.ring_count[:n]), list(lay.ring_step[:n])), row_of

    def rows(self):
        n = lib().jf_grid_rows(C.byref(self.c))
        if n < 0:
            raise JfError(n, lib().jf_last_error(None).decode())
        return n

    def interpolation(self, ele, azi):
        idx = np.zeros(4, np.int32)
        om = np.zeros(6, np.float32)
        rc = lib().jf_grid_interpolation(C.byref(self.c), ele, azi, _ip(idx), _fp(om))
        return None if rc else (idx, om)

    def pick(self, ele, azi):
        return lib().jf_grid_pick(C.byref(self.c), ele, azi)


class Engine:
    """Thin object wrapper; method names follow the C ABI."""

    def __init__(self, B, hrtf_len, n_sources, hrir=None, hrir_dir=None, device=0, max_batch_blocks=1, flags=0, grid=None,
                 sofa=None, sofa_tol_deg=0.05):
        L = lib()
        cfg = JfConfig(B, hrtf_len, n_sources, device, max_batch_blocks, flags)
        h = C.c_void_p()
        if sofa is not None:
            rc = L.jf_engine_create_sofa(C.byref(cfg), os.fsencode(sofa), sofa_tol_deg, C.byref(h))
        elif grid is not None:
            hrir = np.ascontiguousarray(hrir, np.float32)
            assert hrir.ndim == 3 and hrir.shape[0] == grid.rows() and hrir.shape[1] == 2  # the C side reads rows x 2 x taps floats
            self._grid = grid
            rc = L.jf_engine_create_grid(C.byref(cfg), C.byref(grid.c), _fp(hrir), hrir.shape[2], C.byref(h))
        elif hrir_dir is not None:
            rc = L.jf_engine_create_from_dir(C.byref(cfg), hrir_dir.encode(), C.byref(h))
        else:
            hrir = np.ascontiguousarray(hrir, np.float32)
            assert hrir.shape[0] == NUM_HRTF and hrir.shape[1] == 2
            rc = L.jf_engine_create(C.byref(cfg), _fp(hrir), hrir.shape[2], C.byref(h))
        if rc:
            raise JfError(rc, L.jf_last_error(None).decode())
        self.h = h
        self.B, self.S, self.maxK = B, n_sources, max_batch_blocks
        self.N = L.jf_pad_len(h)

    def close(self):
        if getattr(self, "h", None):
            lib().jf_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise JfError(rc, lib().jf_last_error(self.h).decode())

    def set_signal(self, s, mono):
        mono = np.ascontiguousarray(mono, np.float32)
        self._chk(lib().jf_source_set_signal(self.h, s, _fp(mono), len(mono)))

    def set_spherical(self, s, ele, azi, r):
        return lib().jf_source_set_spherical(self.h, s, ele, azi, r)

    def set_cartesian(self, s, x, y, z):
        return lib().jf_source_set_cartesian(self.h, s, x, y, z)

    def get_position(self, s):
        o = np.zeros(6, np.float32)
        self._chk(lib().jf_source_get_position(self.h, s, _fp(o)))
        return o

    def reset(self, s):
        self._chk(lib().jf_source_reset(self.h, s))

    def process_block(self):
        out = np.zeros(2 * self.B, np.float32)
        self._chk(lib().jf_process_block(self.h, _fp(out)))
        return out

    def submit_block(self):
        return lib().jf_submit_block(self.h)

    def collect_block(self):
        out = np.zeros(2 * self.B, np.float32)
        rc = lib().jf_collect_block(self.h, _fp(out))
        return rc, out

    def callback(self):
        out = np.zeros(2 * self.B, np.float32)
        self._chk(lib().jf_callback(self.h, _fp(out)))
        return out

    def set_mode(self, mode):
        self._chk(lib().jf_set_mode(self.h, int(mode)))

    def set_pause(self, p):
        self._chk(lib().jf_set_pause(self.h, int(p)))

    def process_batch(self, pos):
        pos = np.ascontiguousarray(pos, np.float32)
        K, S = pos.shape[0], pos.shape[1]
        assert S == self.S and pos.shape[2] == 5
        mix = np.zeros((K, 2 * self.B), np.float32)
        self._chk(lib().jf_process_batch(self.h, K, _fp(pos), _fp(mix)))
        return mix

    def set_latched(self, records):
        """every source's position := its latched record [S][5] (what S setter calls leave behind)"""
        records = np.ascontiguousarray(records, np.float32)
        assert records.shape == (self.S, 5)
        self._chk(lib().jf_sources_set_latched(self.h, _fp(records)))

    def upload_positions(self, pos):
        pos = np.ascontiguousarray(pos, np.float32)
        assert pos.shape[1] == self.S and pos.shape[2] == 5
        self._chk(lib().jf_batch_upload_positions(self.h, pos.shape[0], _fp(pos)))

    def batch_run(self, first, n, d_out=None):
        self._chk(lib().jf_batch_run(self.h, first, n, d_out))

    def synchronize(self):
        self._chk(lib().jf_synchronize(self.h))

    def batch_fetch(self, n_blocks):
        """the engine's own mix of the last batch_run (d_out_mix = NULL), [n_blocks][2B], on the host"""
        out = np.empty((n_blocks, 2 * self.B), np.float32)
        self._chk(lib().jf_batch_fetch(self.h, int(n_blocks), _fp(out)))
        return out

    def mix_device_ptr(self):
        return lib().jf_batch_mix_device(self.h)

    def partial_device_ptr(self):
        return lib().jf_batch_partial_device(self.h)

    def stream_ptr(self):
        return lib().jf_engine_stream(self.h)

    def profile_enable(self, on):
        self._chk(lib().jf_profile_enable(self.h, int(on)))

    def profile_set_stride(self, every):
        self._chk(lib().jf_profile_set_stride(self.h, int(every)))

    def profile_read(self):
        f, p, m = C.c_double(), C.c_double(), C.c_double()
        n = C.c_long()
        self._chk(lib().jf_profile_read(self.h, C.byref(f), C.byref(p), C.byref(m), C.byref(n)))
        return {"fused_ms": f.value, "prep_ms": p.value, "mix_ms": m.value, "launches": n.value}

    def set_reverb(self, ir, gain=1.0):
        ir = np.ascontiguousarray(ir, np.float32)
        self._chk(lib().jf_reverb_set_ir(self.h, _fp(ir) if len(ir) else None, len(ir), gain))

    def profile_read_reverb(self):
        r = C.c_double()
        self._chk(lib().jf_profile_read_reverb(self.h, C.byref(r)))
        return r.value

    def set_source_group(self, g):
        self._chk(lib().jf_debug_set_source_group(self.h, int(g)))

    def last_kernels(self):
        return lib().jf_debug_last_kernels(self.h).decode().split(";")

    def read_stamps(self, n):
        out = np.zeros(n, np.uint64)
        self._chk(lib().jf_debug_read_stamps(self.h, out.ctypes.data_as(C.POINTER(C.c_ulonglong)), n))
        return out

    def source_order(self):
        o = np.zeros(self.S, np.int32)
        self._chk(lib().jf_debug_source_order(self.h, _ip(o)))
        return o

    def last_source_group(self):
        return lib().jf_debug_last_source_group(self.h)

    def set_grid_limit(self, wgs):
        self._chk(lib().jf_debug_set_grid_limit(self.h, int(wgs)))

    def set_prep_ahead(self, on):
        self._chk(lib().jf_debug_set_prep_ahead(self.h, int(bool(on))))

    def last_block_peak(self):
        return float(lib().jf_last_block_peak(self.h))

    def stage_taps(self, positions, windows=None):
        """positions [n][5] (and windows [n][1024]) -> D [n][513] complex64 (and Y [n][2][513] complex64)."""
        positions = np.ascontiguousarray(positions, np.float32)
        n = positions.shape[0]
        dist = np.zeros((n, NC, 2), np.float32)
        spec = None
        if windows is not None:
            windows = np.ascontiguousarray(windows, np.float32)
            assert windows.shape == (n, PAD_LEN)
            spec = np.zeros((n, 2, NC, 2), np.float32)
        self._chk(lib().jf_debug_stage_taps(self.h, n, _fp(positions), _fp(windows) if spec is not None else None,
                                            _fp(dist), _fp(spec) if spec is not None else None))
        d = dist.view(np.complex64)[..., 0]
        return d if spec is None else (d, spec.view(np.complex64)[..., 0])

    def set_reverb_partitioning(self, how):
        """0 by length, 1 uniform, 2 non-uniform; in effect from the next set_reverb"""
        self._chk(lib().jf_debug_set_reverb_partitioning(self.h, int(how)))

    def set_reverb_async(self, on):
        """one-block calls: the big partitions' kernels on a second stream (default) or in line"""
        self._chk(lib().jf_debug_set_reverb_async(self.h, int(bool(on))))

    def set_reverb_ahead(self, on):
        """one-block calls launch the next block's reverb stage behind their own spatialiser (default) or not"""
        self._chk(lib().jf_debug_set_reverb_ahead(self.h, int(bool(on))))

    def set_reverb_side_workgroups(self, n):
        """workgroups of the product kernel on the reverb's side stream (tuning runs)"""
        self._chk(lib().jf_debug_set_reverb_side_workgroups(self.h, int(n)))

    def set_reverb_lazy_state(self, on):
        """batch calls of whole big blocks put the small transforms of their last blocks off (default) or form them at once"""
        self._chk(lib().jf_debug_set_reverb_lazy_state(self.h, int(bool(on))))

    def set_reverb_head_fused(self, on):
        """one-block calls: the reverb's head inside the real-time kernel's launch, or as a kernel of its own (default: measured 5 us
        faster per block)"""
        self._chk(lib().jf_debug_set_reverb_head_fused(self.h, int(bool(on))))

    def reverb_partitions(self):
        """(partitions of B the response has, head partitions in use, big partitions, taps per big partition)"""
        h, b, t = C.c_int(), C.c_int(), C.c_int()
        n = lib().jf_debug_reverb_partitions(self.h, C.byref(h), C.byref(b), C.byref(t))
        return n, h.value, b.value, t.value

    def set_reverb_form(self, form):
        self._chk(lib().jf_debug_set_reverb_form(self.h, int(form)))

    def read_device(self, ptr, shape):
        out = np.zeros(shape, np.float32)
        self._chk(lib().jf_debug_copy_from_device(self.h, ptr, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def set_rt_max_sources(self, n):
        self._chk(lib().jf_debug_set_rt_max_sources(self.h, int(n)))

    def table_rows(self):
        return lib().jf_table_rows(self.h)

    def read_table(self):
        t = np.zeros((self.table_rows(), 2, NC, 2), np.float32)
        self._chk(lib().jf_debug_read_table(self.h, _fp(t)))
        return t.view(np.complex64)[..., 0]

    def set_interp_table(self, on):
        """0 / False = never, 1 / True = always, 2 = decided per run (default)"""
        self._chk(lib().jf_debug_set_interp_table(self.h, int(on)))

    def last_run_used_rows(self):
        return bool(lib().jf_debug_last_run_used_rows(self.h))

    def interp_table(self):
        """0 = off / not built, 1 = always, 2 = decided per run"""
        return lib().jf_debug_interp_table(self.h)

    def interp_table_built(self):
        """the engine holds the 47 160 pre-interpolated rows (built by the first run that takes them)"""
        return bool(lib().jf_debug_interp_table_built(self.h))

    def count_desc_flags(self, n_items, mask):
        n = lib().jf_debug_count_desc_flags(self.h, int(n_items), int(mask))
        if n < 0:
            self._chk(n)
        return n

    def read_table_rows(self, first_row, n):
        """n rows in the device layout: [n][512][4] = {L.re, L.im, R.re, R.im} (bin 0: {L[0], L[512], R[0], R[512]})."""
        t = np.zeros((n, 512, 4), np.float32)
        self._chk(lib().jf_debug_read_table_rows(self.h, int(first_row), int(n), _fp(t)))
        return t

    def interp_device(self, ele, azi):
        ele = np.ascontiguousarray(ele, np.float32)
        azi = np.ascontiguousarray(azi, np.float32)
        n = len(ele)
        rows = np.zeros((n, 4), np.int32)
        w = np.zeros((n, 4), np.float32)
        nt = np.zeros(n, np.int32)
        self._chk(lib().jf_debug_interp_device(self.h, n, _fp(ele), _fp(azi), _ip(rows), _fp(w), _ip(nt)))
        return rows, w, nt

    def rfft_device(self, windows):
        windows = np.ascontiguousarray(windows, np.float32)
        n = windows.shape[0]
        assert windows.shape[1] == PAD_LEN
        sp = np.zeros((n, NC, 2), np.float32)
        self._chk(lib().jf_debug_rfft_device(self.h, n, _fp(windows), _fp(sp)))
        return sp.view(np.complex64)[..., 0]
