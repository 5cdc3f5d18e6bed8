"""Synthetic multi-source workloads of SURVEY.md 8(d) (BASELINE.json configs 3/4) and
the source sharding used for multi-GPU runs.  Pure NumPy + the C ABI's position helper;
no GPU needed to build a workload.

config 3: per source `src` (GLOBAL id, so shards of a multi-GPU job are disjoint pieces of
one job): 1 s of uniform white noise in [-0.5, 0.5), seed 1234 + src, looped; start azimuth
(src*37) mod 360, elevation -40 + (src*7) mod 121, radius uniform in [0.5, 3.5] from the same
generator; "moving" = azimuth + 1 degree every block (a crossfade every block).
"""
import numpy as np

FS = 44100
TABLE_ROW_BYTES = 2 * 513 * 8   # one HRTF row pair in the reference layout
WINDOW_BYTES = 1024 * 4


def shard_range(n_total, world, rank):
    """Contiguous source range of `rank`: sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def source_signal_and_start(src, n_samples=FS):
    rng = np.random.default_rng(1234 + int(src))
    r = np.float32(rng.uniform(0.5, 3.5))  # drawn first, so it does not depend on n_samples
    sig = rng.uniform(-0.5, 0.5, n_samples).astype(np.float32)
    azi = (int(src) * 37) % 360
    ele = -40 + (int(src) * 7) % 121
    return sig, ele, azi, r


def trajectories(jf, src_ids, n_blocks, moving=True, first_block=0, ele_override=None, move_every=1):
    """Latched position records [n_blocks][len(src_ids)][5] for blocks first_block..
    ele_override: every source at this elevation (tuning runs: 360 distinct positions instead of 43 560).
    move_every: the azimuth advances by one degree every move_every-th block (1: SURVEY.md 8d's "moving";
    172: the dwell of the reference's own benchmarkTesting, precision_test.cu:2093-2152)."""
    src_ids = np.asarray(src_ids)
    ele = np.array([-40 + (int(s) * 7) % 121 for s in src_ids], np.float32)
    if ele_override is not None:
        ele[:] = ele_override
    azi0 = np.array([(int(s) * 37) % 360 for s in src_ids], np.int64)
    r = np.array([source_signal_and_start(s, 1)[3] for s in src_ids], np.float32)
    b = np.arange(first_block, first_block + n_blocks, dtype=np.int64)[:, None]
    azi = (azi0[None, :] + (b // move_every if moving else 0 * b)) % 360
    ele2 = np.broadcast_to(ele[None, :], azi.shape)
    r2 = np.broadcast_to(r[None, :], azi.shape)
    return jf.positions_from_spherical(ele2, azi.astype(np.float32), r2)


def n_terms_table(jf):
    """terms[ele + 49][azi] = number of table rows read for that latched (ele, azi), azi 0..360."""
    t = np.zeros((140, 361), np.int32)
    for e in range(-49, 91):
        for a in range(361):
            idx, _ = jf.interpolation(float(e), float(a))
            if idx[0] == idx[1] == idx[2] == idx[3]:
                n = 1
            elif (idx[0] == idx[2] and idx[1] == idx[3]) or (idx[0] == idx[1] and idx[0] != idx[2]):
                n = 2
            else:
                n = 4
            t[e + 49, a] = n
    return t


def algorithmic_bytes(jf, pos, B, first_old=None, terms=None, pre_rows=False):
    """Algorithmic bytes of a [K][S][5] trajectory window, SURVEY.md 8(d): per source-block
    (rows_old + rows_new) * 8208 B of table + 4096 B window + 2*B*4 B stereo block out.
    first_old: (ele, azi) [S][2] latched before the window (None -> (0, 0), a fresh engine).
    pre_rows: every filter set is one pre-interpolated row (8 KiB; whole-degree positions, include/jefferson.h)."""
    if terms is None:
        terms = n_terms_table(jf)
    ele = pos[..., 0].astype(np.int64)
    azi = pos[..., 1].astype(np.int64)
    K, S = ele.shape
    prev_e = np.zeros((K, S), np.int64)
    prev_a = np.zeros((K, S), np.int64)
    if first_old is not None:
        prev_e[0], prev_a[0] = first_old[:, 0], first_old[:, 1]
    prev_e[1:], prev_a[1:] = ele[:-1], azi[:-1]
    n_new = terms[ele + 49, azi]
    moved = (prev_e != ele) | (prev_a != azi)
    n_old = np.where(moved, terms[prev_e + 49, prev_a], 0)
    if pre_rows:
        n_new, n_old = np.ones_like(n_new), moved.astype(n_new.dtype)
    rows = int(n_new.sum() + n_old.sum())
    items = K * S
    return rows * TABLE_ROW_BYTES + items * (WINDOW_BYTES + 2 * B * 4), rows, items


def algorithmic_bytes_cyclic(jf, pos, blocks_per_step, B, first_step, n_steps, terms=None, pre_rows=False):
    """The same for a run that walks one uploaded period of positions [n_pos][S][5] again and again, one step =
    blocks_per_step consecutive blocks (bench.py): every step of the period is priced once -- the block before
    its first one is the one before it on the circle -- and counted as often as steps first_step ..
    first_step + n_steps - 1 visit it.  n_pos must be a multiple of blocks_per_step."""
    if terms is None:
        terms = n_terms_table(jf)
    n_pos = pos.shape[0]
    assert n_pos % blocks_per_step == 0
    per_step = []
    for j in range(n_pos // blocks_per_step):
        first_old = pos[(j * blocks_per_step - 1) % n_pos, :, :2].astype(np.int64)
        per_step.append(algorithmic_bytes(jf, pos[j * blocks_per_step:(j + 1) * blocks_per_step], B,
                                          first_old=first_old, terms=terms, pre_rows=pre_rows))
    tot = [0, 0, 0]
    for i in range(first_step, first_step + n_steps):
        for c, v in enumerate(per_step[i % len(per_step)]):
            tot[c] += v
    return tuple(tot)



# ---------------------------------------------------------------------------------------------------
# Floating-point work of the path (for the fp32 vector roofline of bench.py).  Textbook counts:
# a complex FFT of n points = 5 n log2 n flops; a complex multiply = 6, a complex add = 2.
N = 1024
NC = 513
FLOPS_RFFT = 5 * 512 * 9 + 512 * 12        # real FFT 1024 = complex FFT 512 + split pass (2 complex add, 1 mul, 1 add per bin)
FLOPS_DISTANCE = NC * 20 + NC * 6          # D[k] (two short polynomials + scaling per bin) and X[k] D[k]
FLOPS_IFFT_FULL = 5 * 1024 * 10            # both ears in one complex 1024-point inverse (the reference: two c2r of 1024)
FLOPS_IFFT_PRUNED = 4 * 5 * 256 * 8        # 4 decimated 256-point transforms ...
FLOPS_XFADE_PER_FRAME = 2 * 3              # out = old (1 - f) + new f, two channels


def flops_filter(n_rows):
    """sum_t w_t H_t (both ears, re and im: 4 (2 n - 1) flops per bin) times X D (two complex multiplies) and the
    two complex adds that form Z = Y_L + j Y_R and its mirror: per bin 8 n + 12."""
    return NC * (8 * n_rows + 12)


def flops_ifft_pruned(B):
    """... + the last radix-4 of the inverse formed for the B frames of the block only (3 complex multiplies and
    3 complex adds per frame)."""
    return FLOPS_IFFT_PRUNED + B * 24


def flops_window(jf, pos, B, G, first_old=None, terms=None, old_sets_spectral=True, pre_rows=False):
    """Floating-point operations of one launch over the trajectory window pos [K][S][5].

    Returns (executed, reference): `executed` is what the shipped kernels have to do -- per source-block one
    forward transform, the distance factor, one weighted filter per set; per UNIT of G consecutive sources one
    pruned inverse for the sum of the new sets and, if any source of the unit crossfades, one for the sum of the
    old sets (G = 1: one or two inverses per source-block; old_sets_spectral=False: one inverse per crossfading
    unit's SOURCE for the old sets, the form of the round-1 group kernel), the crossfade and the G-fold sum -- and `reference`
    is the reference's own algorithm (GPUSoundSource.cu:320-385): an unpruned inverse pair per set and source.
    Silent items (position not interpolable) are counted like the others; the synthetic workloads have none.
    pre_rows: the executed filters read pre-interpolated rows (no weighting: 12 flops per bin and set); `reference` is
    unchanged -- the reference weights four rows per set and block."""
    if terms is None:
        terms = n_terms_table(jf)
    ele = pos[..., 0].astype(np.int64)
    azi = pos[..., 1].astype(np.int64)
    K, S = ele.shape
    prev_e = np.zeros((K, S), np.int64)
    prev_a = np.zeros((K, S), np.int64)
    if first_old is not None:
        prev_e[0], prev_a[0] = first_old[:, 0], first_old[:, 1]
    prev_e[1:], prev_a[1:] = ele[:-1], azi[:-1]
    n_new = terms[ele + 49, azi]
    moved = (prev_e != ele) | (prev_a != azi)
    n_old = np.where(moved, terms[prev_e + 49, prev_a], 0)
    items = K * S
    front = items * (FLOPS_RFFT + FLOPS_DISTANCE)
    filt = int((NC * (8 * n_new + 12)).sum() + (NC * (8 * n_old + 12) * moved).sum())
    filt_ex = int(items * NC * 12 + moved.sum() * NC * 12) if pre_rows else filt
    units_x = int(moved.reshape(K, S // G, G).any(axis=2).sum())     # units with a crossfade
    units = K * (S // G)
    old_inv = units_x if (old_sets_spectral or G == 1) else units_x * G
    inv = (units + old_inv) * flops_ifft_pruned(B)
    xfade = units_x * B * FLOPS_XFADE_PER_FRAME
    gsum = items * NC * 4 * (1 + moved.mean()) if G > 1 else 0          # spectral sums over the unit's sources
    mix = units * 2 * B                                                 # mix_kernel: one add per float of a block
    executed = front + filt_ex + inv + xfade + int(gsum) + mix
    reference = (front + filt + int((1 + moved).sum()) * FLOPS_IFFT_FULL + int(moved.sum()) * B * FLOPS_XFADE_PER_FRAME
                 + items * 2 * B)
    return executed, reference


def flops_cyclic(jf, pos, blocks_per_step, B, G, first_step, n_steps, terms=None, old_sets_spectral=True, pre_rows=False):
    """flops_window over a cyclically walked period of positions (see algorithmic_bytes_cyclic)."""
    if terms is None:
        terms = n_terms_table(jf)
    n_pos = pos.shape[0]
    assert n_pos % blocks_per_step == 0
    per_step = []
    for j in range(n_pos // blocks_per_step):
        first_old = pos[(j * blocks_per_step - 1) % n_pos, :, :2].astype(np.int64)
        per_step.append(flops_window(jf, pos[j * blocks_per_step:(j + 1) * blocks_per_step], B, G,
                                     first_old=first_old, terms=terms, old_sets_spectral=old_sets_spectral,
                                     pre_rows=pre_rows))
    ex = ref = 0
    for i in range(first_step, first_step + n_steps):
        e, r = per_step[i % len(per_step)]
        ex += e
        ref += r
    return ex, ref
