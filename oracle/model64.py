"""float64 NumPy model of the reference's HRTF convolution path.

TEST INFRASTRUCTURE ONLY (same rules as oracle/jf_oracle.h): imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

It evaluates the reference's formulas in float64 while
keeping every quantity the reference holds in a float32 variable (weights,
radius, fsvs, frac, crossfade ramp, HRIR samples, input samples) at its float32
value, so it is "the reference with an exact FFT and exact pointwise maths".
It is the truth the float32 paths (C oracle, HIP kernel) are measured against;
tolerances are stated in the tests.  Parity with the reference binary itself is
unpinned (no golden outputs exist; see jf_oracle.h).

The index/weight functions are a second, independent restatement of
SoundSource.cu:65-105 and hrtf_signals.cu:20-51 in float32 NumPy scalars; the
tests require them to agree bit-for-bit with the C oracle.

Citations are relative to /root/reference/Jefferson/src/.
"""
import math

import numpy as np

f32 = np.float32
PI = 3.14159265358979323846264338327950288  # Universal.cuh:14-16

NUM_ELEV = 14
NUM_HRTF = 710
ELEVATION_POS = [-40, -30, -20, -10, 0, 10, 20, 30, 40, 50, 60, 70, 80, 90]
AZIMUTH_INC = [f32(v) for v in (6.43, 6.00, 5.00, 5.00, 5.00, 5.00, 5.00, 6.00,
                                6.43, 8.00, 10.00, 15.00, 30.00, 361.0)]


def _c_round(x):
    """std::round on a float: half away from zero."""
    x = float(x)
    return f32(math.copysign(math.floor(abs(x) + 0.5), x))


def _c_div(a, b):
    """C integer division (truncates toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def azimuth_offsets():
    """hrtf_signals.cu:119-140."""
    off = [0]
    j = 0
    for i in range(NUM_ELEV):
        azi = f32(0)
        while azi < 360:
            j += 1
            azi = f32(azi + AZIMUTH_INC[i])
        off.append(j)
    return off


AZIMUTH_OFFSET = azimuth_offsets()


def table_positions():
    """(elevation, (int)round(azi)) for each of the 710 rows (hrtf_signals.cu:121-124)."""
    out = []
    for i in range(NUM_ELEV):
        azi = f32(0)
        while azi < 360:
            out.append((ELEVATION_POS[i], int(_c_round(azi))))
            azi = f32(azi + AZIMUTH_INC[i])
    return out


def pick_hrtf(obj_ele, obj_azi):
    """hrtf_signals.cu:20-51."""
    obj_ele = f32(_c_round(f32(obj_ele) / f32(10)) * f32(10))
    dmin = f32(1e37)
    ele_idx = 0
    for i in range(NUM_ELEV):
        d = f32(obj_ele - f32(ELEVATION_POS[i]))
        d = d if d > 0 else -d
        if d < dmin:
            dmin, ele_idx = d, i
    obj_azi = _c_round(obj_azi)
    dmin = f32(1e37)
    hrtf_idx = 0
    n = AZIMUTH_OFFSET[ele_idx + 1] - AZIMUTH_OFFSET[ele_idx]
    for i in range(n):
        d = f32(obj_azi - f32(f32(i) * AZIMUTH_INC[ele_idx]))
        d = d if d > 0 else -d
        if d < dmin:
            dmin, hrtf_idx = d, AZIMUTH_OFFSET[ele_idx] + i
    return hrtf_idx


def interp(ele, azi):
    """SoundSource.cu:65-105 -> (idx[4], omegas[6] float32) or None (missing ring)."""
    ele, azi = f32(ele), f32(azi)
    phi0 = _c_div(int(ele), 10) * 10
    phi1 = _c_div(int(f32(ele + f32(9))), 10) * 10
    omE = f32(f32(ele - f32(phi0)) / f32(10))
    omF = f32(f32(f32(phi1) - ele) / f32(10))
    dt1 = dt2 = None
    for i in range(NUM_ELEV):
        if phi0 == ELEVATION_POS[i]:
            dt1 = AZIMUTH_INC[i]
        if phi1 == ELEVATION_POS[i]:
            dt2 = AZIMUTH_INC[i]
            break
    if dt1 is None or dt2 is None:
        return None

    def lo(dt):
        return int(f32(f32(int(f32(azi / dt))) * dt))

    def hi(dt):
        return int(f32(f32(int(f32(f32(f32(azi + dt) - f32(1)) / dt))) * dt))

    th = [lo(dt1), hi(dt1), lo(dt2), hi(dt2)]
    omA = f32(f32(azi - f32(th[0])) / dt1)
    omB = f32(f32(f32(th[1]) - azi) / dt1)
    omC = f32(f32(azi - f32(th[2])) / dt2)
    omD = f32(f32(f32(th[3]) - azi) / dt2)
    idx = [pick_hrtf(phi0, th[0]), pick_hrtf(phi0, th[1]),
           pick_hrtf(phi1, th[2]), pick_hrtf(phi1, th[3])]
    return idx, [omA, omB, omC, omD, omE, omF]


def interp_corrected(ele, azi):
    """The corrected rule behind JF_FLAG_CORRECTED_INTERPOLATION (not in the reference):
    true floor of the elevation, azimuth folded into [0, 360) with a ring's last interval wrapping to its first
    entry, float azimuths (a ring's weights sum to 1), elevations below the lowest ring clamped to it.
    float32 step by step, like jfo_interp_corrected."""
    ele, azi = f32(ele), f32(azi)
    if not (ele <= 90) or not (ele > -1e6) or not (-1e6 < azi < 1e6):
        return None
    if ele < -40:
        ele = f32(-40)
    a = f32(azi - f32(f32(360) * f32(math.floor(f32(azi / f32(360))))))
    if not (a < 360):
        a = f32(0)
    q = f32(math.floor(f32(ele / f32(10))))
    phi0 = f32(f32(10) * q)
    on_ring = bool(ele == phi0)
    r0 = int(q) + 4
    r1 = r0 if on_ring else r0 + 1
    omE = f32(0) if on_ring else f32(f32(ele - phi0) / f32(10))
    idx, om = [], []
    for r in (r0, r1):
        d = AZIMUTH_INC[r]
        n = AZIMUTH_OFFSET[r + 1] - AZIMUTH_OFFSET[r]
        i0 = min(int(math.floor(f32(a / d))), n - 1)
        wa = f32(f32(a - f32(f32(i0) * d)) / d)
        wa = f32(min(max(wa, f32(0)), f32(1)))
        if n == 1:
            wa = f32(0)
        i1 = 0 if i0 + 1 == n else i0 + 1
        if wa == 0:
            i1 = i0
        idx += [AZIMUTH_OFFSET[r] + i0, AZIMUTH_OFFSET[r] + i1]
        om += [wa, f32(f32(1) - wa)]
    om += [omE, f32(f32(1) - omE)]
    return idx, om


class Grid:
    """Any grid of elevation rings (jf_oracle.h: jfo_grid_interp; the drop-in's jf_engine_create_grid): ring r at ele[r] degrees
    (ascending) with count[r] measurements at azimuths i * step[r] (step None: 360 / count); rows ring by ring."""

    def __init__(self, ring_ele, ring_count, ring_step=None):
        self.ele = [f32(v) for v in ring_ele]
        self.count = [int(v) for v in ring_count]
        self.step = [f32(f32(360) / f32(n)) if ring_step is None else f32(ring_step[r]) for r, n in enumerate(self.count)]
        for r, n in enumerate(self.count):
            if n == 1 and not self.step[r] >= 360:
                self.step[r] = f32(361)
        self.offset = [0]
        for n in self.count:
            self.offset.append(self.offset[-1] + n)
        self.n_rows = self.offset[-1]

    @staticmethod
    def kemar():
        return Grid(ELEVATION_POS, [AZIMUTH_OFFSET[i + 1] - AZIMUTH_OFFSET[i] for i in range(NUM_ELEV)], AZIMUTH_INC)

    def interp(self, ele, azi):
        """The corrected rule in its general form; float32 step by step like jfo_grid_interp."""
        ele, azi = f32(ele), f32(azi)
        if not (ele <= 90) or not (ele > -1e6) or not (-1e6 < azi < 1e6):
            return None
        last = len(self.ele) - 1
        ele = max(ele, self.ele[0])
        ele = min(ele, self.ele[last])
        a = f32(azi - f32(f32(360) * f32(np.floor(f32(azi / f32(360))))))
        if not a < 360:
            a = f32(0)
        r0 = 0
        while r0 < last and self.ele[r0 + 1] <= ele:
            r0 += 1
        phi0 = self.ele[r0]
        on_ring = bool(ele == phi0)
        omE = f32(0) if on_ring else f32(f32(ele - phi0) / f32(self.ele[r0 + 1] - phi0))
        idx, om = [], []
        for r in (r0, r0 if on_ring else r0 + 1):
            d = self.step[r]
            n = self.count[r]
            i0 = min(int(np.floor(f32(a / d))), n - 1)
            wa = f32(f32(a - f32(f32(i0) * d)) / d)
            wa = min(max(wa, f32(0)), f32(1))
            if n == 1:
                wa = f32(0)
            i1 = 0 if i0 + 1 == n else i0 + 1
            if wa == 0:
                i1 = i0
            idx += [self.offset[r] + i0, self.offset[r] + i1]
            om += [wa, f32(f32(1) - wa)]
        return idx, om + [omE, f32(f32(1) - omE)]

    def pick(self, ele, azi):
        ele, azi = f32(ele), f32(azi)
        ring, dmin = 0, f32(1e37)
        for r, e in enumerate(self.ele):
            d = f32(ele - e)
            d = d if d > 0 else -d
            if d < dmin:
                dmin, ring = d, r
        a = f32(azi - f32(f32(360) * f32(np.floor(f32(azi / f32(360))))))
        if not a < 360:
            a = f32(0)
        i = int(np.floor(f32(f32(a / self.step[ring]) + f32(0.5))))
        if i >= self.count[ring]:
            i = 0
        return self.offset[ring] + i


def case_of(h):
    """GPUSoundSource.cu:301-316."""
    if h[0] == h[1] == h[2] == h[3]:
        return 1
    if h[0] == h[2] and h[1] == h[3]:
        return 2
    if h[0] == h[1] and h[0] != h[2]:
        return 3
    return 4


def terms(h, om):
    """(row, weight float32) list in accumulation order (GPUSoundSource.cu:118-292)."""
    c = case_of(h)
    if c == 1:
        return [(h[0], f32(1))]
    if c == 2:
        return [(h[0], om[1]), (h[1], om[0])]
    if c == 3:
        return [(h[0], om[5]), (h[2], om[4])]
    return [(h[0], f32(om[5] * om[1])), (h[1], f32(om[5] * om[0])),
            (h[2], f32(om[4] * om[3])), (h[3], f32(om[4] * om[2]))]


def from_spherical(ele, azi, r):
    """SoundSource.cu:41-54 -> (ele, azi, (x, y, z)) all float32."""
    ele, azi, r = _c_round(ele), _c_round(azi), f32(r)
    x = f32(float(r) * math.sin(float(azi) * PI / 180.0))
    z = f32(float(r) * -math.cos(float(azi) * PI / 180.0))
    y = f32(float(r) * math.sin(float(ele) * PI / 180.0))
    return ele, azi, (x, y, z)


def from_cartesian(x, y, z):
    """SoundSource.cu:20-36 -> (ele, azi, r) float32, or None when r == 0."""
    x, y, z = f32(x), f32(y), f32(z)
    r = np.sqrt(f32(f32(f32(x * x) + f32(z * z)) + f32(y * y)))
    hr = np.sqrt(f32(f32(x * x) + f32(z * z)))
    if r == 0:
        return None
    ele = f32(float(f32(np.arctan2(y, hr) * f32(180.0))) / PI)
    azi = f32(float(f32(np.arctan2(f32(-x / r), f32(-z / r)) * f32(180.0))) / PI)
    if azi < 0:
        azi = f32(azi + f32(360))
    return _c_round(ele), _c_round(azi), f32(r)


def distance_params(coords):
    """GPUSoundSource.cu:81-90: (r' , fsvs, frac) as float32."""
    x, y, z = (f32(c) for c in coords)
    r = np.sqrt(f32(f32(f32(x * x) + f32(y * y)) + f32(z * z)))
    r = f32(r / f32(5))
    fsvs = f32(44100.0 / 343.0)
    frac = f32(f32(1) + f32(fsvs * f32(float(r) ** 2)))
    return r, fsvs, frac


def distance_factor(coords, nc):
    """kernels.cu:116-125 in float64 (complex128[nc])."""
    r, fsvs, frac = distance_params(coords)
    i = np.arange(nc, dtype=np.float64)
    ph = 2 * PI * float(fsvs) * float(r) * i / nc
    return (np.cos(ph) - 1j * np.sin(ph)) / float(frac)


def build_table(hrir, N):
    """hrtf_signals.cu:107-153: unnormalised r2c of zero-padded HRIRs, float64.
    hrir float32 [n][2][taps] -> complex128 [n][2][N/2+1]."""
    return np.fft.rfft(hrir.astype(np.float64), n=N, axis=-1)


class Source:
    def __init__(self, N):
        self.buf = np.zeros(0, np.float32)
        self.count = 0
        self.x = np.zeros(N, np.float64)
        # SoundSource.cu:3-16
        self.ele, self.azi, self.r = f32(0), f32(0), f32(0.5)
        self.coords = (f32(0), f32(0), f32(0.5))
        self.old_ele, self.old_azi = f32(0), f32(0)
        self.last = None


class Model:
    """Data + sources + callback_func (Audio.cu:94-163, CPU timing: zero latency)."""

    def __init__(self, frames_per_buffer, hrtf_len, n_sources, hrir, grid=None):
        """grid: a Grid of the set's own (its rule and pick then replace KEMAR's in every mode); None: the reference's."""
        self.B = int(frames_per_buffer)
        self.N = int(2 ** math.ceil(math.log2(self.B + hrtf_len - 1)))  # Universal.cuh:12
        self.Nc = self.N // 2 + 1
        self.grid = grid
        assert hrir.shape[0] == (NUM_HRTF if grid is None else grid.n_rows) and hrir.shape[1] == 2
        self.table = build_table(np.asarray(hrir, np.float32), self.N)
        self.src = [Source(self.N) for _ in range(n_sources)]
        i = np.arange(self.B, dtype=np.float32)
        fn = (i / f32(self.B - 1.0)).astype(np.float32)  # kernels.cu:134
        self.fade_new = fn.astype(np.float64)
        self.fade_old = (f32(1.0) - fn).astype(np.float64)
        self.mode = 0  # bit 0: 0 = FD_COMPLEX, 1 = FD_BASIC (nearest HRTF, CPUSoundSource.cpp:113-142); bit 1: corrected rule

    def set_signal(self, s, mono):
        self.src[s].buf = np.asarray(mono, np.float32).copy()
        self.src[s].count = 0

    def set_spherical(self, s, ele, azi, r):
        q = self.src[s]
        q.ele, q.azi, q.coords = from_spherical(ele, azi, r)
        q.r = f32(r)

    def set_cartesian(self, s, x, y, z):
        q = self.src[s]
        q.ele, q.azi, q.r = from_cartesian(x, y, z)
        q.coords = (f32(x), f32(y), f32(z))

    def reset(self, s):
        q = self.src[s]
        q.x[:] = 0
        q.count = 0
        q.old_ele, q.old_azi = f32(0), f32(0)

    def _filter(self, X, D, h, om):
        Y = np.zeros((2, self.Nc), np.complex128)
        for row, w in terms(h, om):
            Y += float(w) * (X[None, :] * self.table[row]) * D[None, :]
        Y[:, 0] = Y[:, 0].real
        Y[:, -1] = Y[:, -1].real
        return np.fft.irfft(Y, n=self.N, axis=-1) * self.N  # unnormalised c2r

    def source_block(self, q, ele, azi, coords):
        N, B = self.N, self.B
        L = len(q.buf)
        if L == 0:
            new = np.zeros(B)
        else:
            pos = (q.count + np.arange(B)) % L
            new = q.buf[pos].astype(np.float64)
            q.count = int((q.count + B) % L)
        q.x[N - B:] = new
        X = np.fft.rfft(q.x) / N
        if self.mode & 1:
            Y = X[None, :] * self.table[pick_hrtf(ele, azi) if self.grid is None else self.grid.pick(ele, azi)]
            Y[:, 0] = Y[:, 0].real
            Y[:, -1] = Y[:, -1].real
            y = (np.fft.irfft(Y, n=N, axis=-1) * N)[:, N - B:]
            q.old_azi, q.old_ele = f32(azi), f32(ele)
            q.x[:N - B] = q.x[B:].copy()
            q.last = y.T.copy().reshape(-1)
            return q.last
        rule = self.grid.interp if self.grid is not None else interp_corrected if self.mode & 2 else interp
        cur = rule(ele, azi)
        xfade = (q.old_azi != azi) or (q.old_ele != ele)
        old = rule(q.old_ele, q.old_azi) if xfade else None
        if cur is None or (xfade and old is None):
            blk = np.zeros((B, 2))
        else:
            D = distance_factor(coords, self.Nc)
            if not xfade:
                y = self._filter(X, D, *cur)[:, N - B:]
            else:
                y1 = self._filter(X, D, *old)[:, N - B:]
                y2 = self._filter(X, D, *cur)[:, N - B:]
                y = y1 * self.fade_old[None, :] + y2 * self.fade_new[None, :]
            blk = y.T.copy()  # [B][2] interleaved
        q.old_azi, q.old_ele = f32(azi), f32(ele)
        q.x[:N - B] = q.x[B:].copy()
        q.last = blk.reshape(-1)
        return q.last

    def process_block(self):
        out = np.zeros(2 * self.B)
        for q in self.src:
            out += self.source_block(q, q.ele, q.azi, q.coords)
        return out

    def process_batch(self, pos):
        """pos float32 [K][S][5] = latched {ele, azi, x, y, z}; -> mix [K][2B], partial [S][K][2B]."""
        K, S = pos.shape[0], pos.shape[1]
        partial = np.zeros((S, K, 2 * self.B))
        for s, q in enumerate(self.src):
            for b in range(K):
                p = pos[b, s]
                q.ele, q.azi, q.coords = f32(p[0]), f32(p[1]), (f32(p[2]), f32(p[3]), f32(p[4]))
                partial[s, b] = self.source_block(q, q.ele, q.azi, q.coords)
        return partial.sum(axis=0), partial
