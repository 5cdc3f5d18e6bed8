"""jf_engine_create_grid on a real MI355X (other HRTF sets; FuturePlans.md:21):

* the reference's own grid through the new entry point IS jf_engine_create: same blocks bit for bit in every mode and
  through every kernel (batch, pair, real-time);
* a synthetic uniform 5 x 10 degree grid and an irregular one (rings 15 degrees apart, the lowest below KEMAR's range,
  counts thinning out to a single measurement at the pole): the kernels' index/weight rule against the C oracle bit for
  bit on dense positions, the rendered blocks against the float32 C oracle (4e-7) and the float64 model (2e-7) per
  source, through the per-source kernel, the pair kernel (with and without pre-interpolated rows), the one-launch
  real-time kernel and FD_BASIC.
"""
import numpy as np
import pytest

import model64
import oracle_lib
from conftest import assert_within, sum_tol
from test_grid import irregular_grid, uniform_grid

pytestmark = pytest.mark.gpu

TOL64 = 2e-7
TOL32 = 4e-7


def _synthetic_hrirs(n_rows, taps=128, seed=21):
    """decaying noise with a per-row delay and gain: rows differ audibly, |H| of order 1"""
    rng = np.random.default_rng(seed)
    h = rng.standard_normal((n_rows, 2, taps)) * np.exp(-np.arange(taps) / 12.0)
    h *= 0.35 / np.sqrt((h ** 2).sum(axis=-1, keepdims=True))
    for j in range(n_rows):
        for ear in range(2):
            h[j, ear] = np.roll(h[j, ear], (j * (ear + 1)) % 9)
    return h.astype(np.float32)


def _trajectory(jf, S, K, lo, hi):
    """whole-degree positions over [lo, hi] x [0, 360): sources that stay, step by a degree, jump; records as the
    spherical setter latches them"""
    pos = np.zeros((K, S, 5), np.float32)
    for s in range(S):
        e0 = lo + (11 * s) % (hi - lo + 1)
        a0 = (47 * s) % 360
        for k in range(K):
            kind = s % 4
            ele = e0 if kind < 2 else lo + (e0 - lo + 9 * k) % (hi - lo + 1)
            azi = a0 if kind == 0 else (a0 + k) % 360 if kind == 1 else (a0 + 40 * k) % 360
            pos[k, s] = jf.position_from_spherical(float(ele), float(azi), 0.3 + 0.05 * s)
            pos[k, s, 0] = ele      # (the helper rounds; the record may lie outside (-50, 90] for a grid of its own)
    return pos


def test_kemars_grid_through_the_new_entry_point_is_jf_engine_create(jf, hrir, castanets):
    S, K, B = 8, 10, 256
    pos = _trajectory(jf, S, K, -40, 90)
    sigs = [(0.5 * np.roll(castanets, 3001 * s)[:9000 + 97 * s]).astype(np.float32) for s in range(S)]
    outs = []
    for grid in (None, jf.Grid.kemar()):
        res = []
        for flags in (0, jf.JF_FLAG_CORRECTED_INTERPOLATION):
            for group in (1, 4):
                e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=K, flags=flags, grid=grid)
                assert e.table_rows() == 710
                e.set_source_group(group)
                for s in range(S):
                    e.set_signal(s, sigs[s])
                res.append(e.process_batch(pos[:6]))
                e.set_mode(jf.JF_MODE_FD_BASIC)
                res.append(e.process_batch(pos[6:8]))
                e.set_mode(jf.JF_MODE_FD_COMPLEX)
                for k in (8, 9):            # the one-launch kernel
                    for s in range(S):
                        assert e.set_spherical(s, pos[k, s, 0], pos[k, s, 1], 0.3 + 0.05 * s) == 0
                    res.append(e.process_block()[None])
                assert e.set_spherical(0, -60.0, 0.0, 1.0) == jf.JF_ERR_RANGE      # the reference's range either way
                e.close()
        outs.append(np.concatenate(res))
    assert np.abs(outs[0]).max() > 0.05
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("which", ["uniform", "irregular"])
def test_rule_on_the_device_equals_the_oracle(jf, which):
    ele, cnt, step = uniform_grid() if which == "uniform" else irregular_grid()
    g, og = jf.Grid(ele, cnt, step), oracle_lib.Grid(ele, cnt, step)
    e = jf.Engine(256, 512, 1, hrir=_synthetic_hrirs(g.rows()), grid=g)
    assert e.table_rows() == og.n_rows
    rng = np.random.default_rng(8)
    pe = np.concatenate([np.repeat(np.arange(-92, 93, 1.0), 40), rng.uniform(-95, 95, 6000)]).astype(np.float32)
    pa = np.concatenate([np.tile(np.arange(-4, 396, 10.0), 185), rng.uniform(-30, 750, 6000)]).astype(np.float32)
    rows, w, nt = e.interp_device(pe, pa)
    e.close()
    bad = 0
    for i in range(len(pe)):
        r = og.interp(float(pe[i]), float(pa[i]))
        if r is None:
            bad += nt[i] != 0
            continue
        orows, ow = oracle_lib.terms(*r)
        n = len(orows)
        bad += not (nt[i] == n and np.array_equal(rows[i, :n], orows) and np.array_equal(w[i, :n], ow))
    assert bad == 0


@pytest.mark.parametrize("which,B", [("uniform", 256), ("irregular", 256), ("irregular", 128)])
def test_blocks_on_a_grid_of_its_own_against_both_oracles(jf, castanets, which, B):
    ele, cnt, step = uniform_grid() if which == "uniform" else irregular_grid()
    g, og, mg = jf.Grid(ele, cnt, step), oracle_lib.Grid(ele, cnt, step), model64.Grid(ele, cnt, step)
    h = _synthetic_hrirs(g.rows())
    S, K = 8, 12
    lo = -60 if which == "irregular" else -40          # below the irregular grid's lowest ring too: clamped
    pos = _trajectory(jf, S, K, lo, 90)
    pos[:, 3, 0] += 0.5                                  # fractional elevations and azimuths for two sources
    pos[:, 6, 1] += 0.25
    sigs = [(0.45 * np.roll(castanets, 2003 * s)[:9000 + 97 * s]).astype(np.float32) for s in range(S)]
    ora = oracle_lib.Engine(B, 512, S, h, grid=og)
    mod = model64.Model(B, 512, S, h, grid=mg)
    for s in range(S):
        ora.set_signal(s, sigs[s])
        mod.set_signal(s, sigs[s])
    _, w32a = ora.process_batch(pos[:8], want_partial=True)
    _, w64a = mod.process_batch(pos[:8])
    ora.set_mode(1)
    mod.mode = 1
    _, w32b = ora.process_batch(pos[8:], want_partial=True)
    _, w64b = mod.process_batch(pos[8:])
    ora.close()
    want32 = np.concatenate([w32a, w32b], axis=1)        # [S][K][2B]
    want64 = np.concatenate([w64a, w64b], axis=1)
    assert 0.05 < np.abs(want64).max() < 1.0

    # per-source kernel: every source's own blocks
    e = jf.Engine(B, 512, S, hrir=h, max_batch_blocks=8, grid=g)
    e.set_source_group(1)
    for s in range(S):
        e.set_signal(s, sigs[s])
    e.upload_positions(pos)
    e.batch_run(0, 8)
    e.synchronize()
    part = [e.read_device(e.partial_device_ptr(), (8, S, 2 * B))]
    e.set_mode(jf.JF_MODE_FD_BASIC)
    e.batch_run(8, 4)
    e.synchronize()
    part.append(e.read_device(e.partial_device_ptr(), (4, S, 2 * B)))
    e.close()
    part = np.concatenate(part).transpose(1, 0, 2)
    for s in range(S):
        assert_within(part[s], want64[s], TOL64, f"grid {which} B={B}: source {s} vs model64")
        assert_within(part[s], want32[s], TOL32, f"grid {which} B={B}: source {s} vs oracle32")

    # pair kernel with and without pre-interpolated rows, and the one-launch kernel: the mix
    mix64 = want64.sum(axis=0)
    for rows in (0, 1):
        e = jf.Engine(B, 512, S, hrir=h, max_batch_blocks=8, grid=g)
        e.set_source_group(4)
        e.set_interp_table(rows)
        for s in range(S):
            e.set_signal(s, sigs[s])
        a = e.process_batch(pos[:8])
        assert e.last_run_used_rows() == bool(rows)
        e.set_mode(jf.JF_MODE_FD_BASIC)
        b = e.process_batch(pos[8:])
        e.close()
        assert_within(np.concatenate([a, b]), mix64, sum_tol(TOL64, S), f"grid {which} B={B}: pair kernel rows={rows} vs model64")
    e = jf.Engine(B, 512, S, hrir=h, grid=g)
    for s in range(S):
        e.set_signal(s, sigs[s])
    got = []
    for k in range(K):
        if k == 8:
            e.set_mode(jf.JF_MODE_FD_BASIC)
        e.set_latched(pos[k])
        got.append(e.process_block())
    assert any("rt_block_kernel" in x for x in e.last_kernels())
    # the setters of an engine with a grid of its own take the whole sphere
    assert e.set_spherical(0, -75.0, 10.0, 1.0) == 0 and e.set_spherical(0, 91.0, 10.0, 1.0) == jf.JF_ERR_RANGE
    e.close()
    assert_within(np.array(got), mix64, sum_tol(TOL64, S), f"grid {which} B={B}: real-time kernel vs model64")


def test_a_group_on_a_grid_of_its_own_and_rings_from_directions(jf, castanets):
    """jf_group_create_grid (one GPU: a communicator of size 1) renders what jf_engine_create_grid renders; and an engine whose
    grid and row order come from the measurements' directions (jf_grid_from_positions: a shuffled SOFA-style list) renders what
    the engine created from the ring description renders, bit for bit."""
    import importlib
    grp = importlib.import_module("jefferson_amd.group")       # needs libjefferson_group.so (RCCL at build time)
    ele, cnt, _ = irregular_grid()
    g = jf.Grid(ele, cnt)
    h = _synthetic_hrirs(g.rows())
    S, K, B = 6, 6, 256
    pos = _trajectory(jf, S, K, -60, 90)
    sigs = [(0.45 * np.roll(castanets, 2003 * s)[:9000 + 97 * s]).astype(np.float32) for s in range(S)]
    e = jf.Engine(B, 512, S, hrir=h, max_batch_blocks=K, grid=g)
    for s in range(S):
        e.set_signal(s, sigs[s])
    want = e.process_batch(pos)
    e.close()
    G = grp.Group(B, 512, S, h, n_gpus=1, max_batch_blocks=K, grid=g)
    for s in range(S):
        G.set_signal(s, sigs[s])
    got = G.process_batch(pos)
    G.close()
    assert np.abs(want).max() > 0.02 and np.array_equal(got, want)
    # the same set as a shuffled list of directions
    rng = np.random.default_rng(12)
    el, az = [], []
    for r, n in enumerate(cnt):
        for i in range(n):
            el.append(float(ele[r]))
            az.append(i * 360.0 / n)
    perm = rng.permutation(len(el))
    g2, row_of = jf.Grid.from_positions(np.array(az, np.float32)[perm], np.array(el, np.float32)[perm], tol_deg=0.01)
    h2 = np.zeros_like(h)
    h2[row_of] = h[perm]              # measurement i (= row perm[i] of the ring-ordered table) goes to row row_of[i]
    assert np.array_equal(h2, h)
    e2 = jf.Engine(B, 512, S, hrir=h2, max_batch_blocks=K, grid=g2)
    for s in range(S):
        e2.set_signal(s, sigs[s])
    got2 = e2.process_batch(pos)
    e2.close()
    assert np.array_equal(got2, want)
