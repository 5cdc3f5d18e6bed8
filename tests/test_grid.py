"""Any grid of elevation rings (include/jefferson.h: jf_hrtf_grid; SURVEY.md 8(f)-2 "other HRTF sets", the reference's
TODO FuturePlans.md:21) -- host side, no GPU: the oracle's general rule is the corrected rule on KEMAR's grid bit for bit,
the product's host twin equals the oracle on KEMAR, on a uniform 5 x 10 degree grid and on an irregular one, and bad
grids are refused."""
import numpy as np
import pytest

import model64
import oracle_lib

KEMAR_COUNTS = [56, 60, 72, 72, 72, 72, 72, 60, 56, 45, 36, 24, 12, 1]      # hrtf_signals.cu:10


def uniform_grid():
    """14 rings at -40 .. 90 in steps of 10, 72 measurements each (5 degrees)"""
    return list(range(-40, 91, 10)), [72] * 14, None


def irregular_grid():
    """rings 15 degrees apart from -45 to 90 with counts that thin out towards the pole, the pole a single measurement;
    the lowest ring below KEMAR's range"""
    return [-45, -30, -15, 0, 15, 30, 45, 60, 75, 90], [24, 30, 36, 40, 36, 30, 24, 12, 7, 1], None


def _positions(lo=-95.0, hi=95.0):
    rng = np.random.default_rng(3)
    whole = [(float(e), float(a)) for e in range(int(lo), int(hi) + 1, 3) for a in range(-10, 372, 7)]
    frac = list(zip(rng.uniform(lo, hi, 3000).astype(np.float32).tolist(),
                    rng.uniform(-20, 740, 3000).astype(np.float32).tolist()))
    edge = [(e, a) for e in (-90.0, -45.0, -40.0, -39.999996, 0.0, 29.999998, 30.0, 89.99999, 90.0, 90.000008)
            for a in (0.0, 359.99997, 360.0, 354.99997, 355.0, 357.5, 6.43, 353.65, 720.0, -0.25)]
    return whole + frac + edge


def test_the_general_rule_on_kemars_grid_is_the_corrected_rule():
    g = oracle_lib.Grid.kemar()
    assert g.count.tolist() == KEMAR_COUNTS and g.n_rows == 710
    m = model64.Grid.kemar()
    n = 0
    for ele, azi in _positions():
        a = oracle_lib.interp(ele, azi, corrected=True)
        b = g.interp(ele, azi)
        c = m.interp(ele, azi)
        assert (a is None) == (b is None) == (c is None), (ele, azi)
        if a is None:
            continue
        n += 1
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (ele, azi, a, b)
        assert list(a[0]) == list(c[0]) and [np.float32(v) for v in c[1]] == list(a[1]), (ele, azi, a, c)
    assert n > 5000


@pytest.mark.parametrize("which", ["kemar", "uniform", "irregular"])
def test_product_host_rule_equals_the_oracle(which):
    from jf_load import jf
    if which == "kemar":
        og, pg, mg = oracle_lib.Grid.kemar(), jf.Grid.kemar(), model64.Grid.kemar()
    else:
        ele, cnt, step = uniform_grid() if which == "uniform" else irregular_grid()
        og, pg, mg = oracle_lib.Grid(ele, cnt, step), jf.Grid(ele, cnt, step), model64.Grid(ele, cnt, step)
    assert pg.rows() == og.n_rows == mg.n_rows
    seen_rows = set()
    for k, (ele, azi) in enumerate(_positions()):
        a, b = og.interp(ele, azi), pg.interpolation(ele, azi)
        assert (a is None) == (b is None), (ele, azi)
        if a is None:
            assert not ele <= 90.0
            continue
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (ele, azi, a, b)
        idx, om = a
        assert ((0 <= idx) & (idx < og.n_rows)).all()
        # weights of a ring sum to 1 (to rounding), the two rings' weights too
        assert abs(float(om[0]) + float(om[1]) - 1) < 2e-7 and abs(float(om[4]) + float(om[5]) - 1) < 2e-7
        if which == "kemar":    # the reference's grid keeps the reference's own search (hrtf_signals.cu:20-51: no wrap at 360)
            assert pg.pick(ele, azi) == oracle_lib.pick_hrtf(ele, azi), (ele, azi)
            assert og.pick(ele, azi) == mg.pick(ele, azi), (ele, azi)
        else:
            assert pg.pick(ele, azi) == og.pick(ele, azi) == mg.pick(ele, azi), (ele, azi)
        seen_rows.update(idx.tolist())
        if k % 7 == 0:      # the float32 NumPy restatement as well (slow)
            c = mg.interp(ele, azi)
            assert list(idx) == list(c[0]) and [np.float32(v) for v in c[1]] == list(om), (ele, azi)
    assert len(seen_rows) > 0.5 * og.n_rows


def test_a_measured_position_is_its_own_row():
    """At a measurement the rule names that one row (all four indices equal -> one term, weight 1: jfo_case 1), and the
    nearest-measurement pick agrees."""
    ele, cnt, _ = irregular_grid()
    g = oracle_lib.Grid(ele, cnt)
    off = np.concatenate([[0], np.cumsum(cnt)])
    for r in range(len(ele)):
        for i in range(0, cnt[r], 5):
            azi = float(np.float32(i) * (np.float32(360) / np.float32(cnt[r])))
            idx, om = g.interp(float(ele[r]), azi)
            rows, w = oracle_lib.terms(idx, om)
            assert om[4] == 0 and idx[0] == idx[2] == off[r] + i
            if om[0] == 0:       # i * step reproduced exactly: one row
                assert list(rows) == [off[r] + i] and list(w) == [1.0]
            assert g.pick(float(ele[r]), azi) == off[r] + i


def test_bad_grids_are_refused():
    from jf_load import jf
    ok = jf.Grid([-10, 0, 10], [8, 12, 8])
    assert ok.rows() == 28
    for ele, cnt, step in (([0, 0], [4, 4], None),              # not ascending
                           ([10, 0], [4, 4], None),
                           ([0, 95], [4, 4], None),             # outside [-90, 90]
                           ([0], [0], None),                    # an empty ring
                           ([0], [4], [60.0]),                  # four steps of 60 do not go round
                           ([0], [4], [120.0]),                 # the fourth measurement would lie at 360
                           ([0], [4], [-90.0]),
                           (list(range(-90, 91, 4)), [4] * 46, None)):      # more than JF_MAX_RINGS rings
        g = jf.Grid(ele, cnt, step)
        with pytest.raises(jf.JfError) as ex:
            g.rows()
        assert ex.value.code == jf.JF_ERR_ARG
        assert g.interpolation(0.0, 0.0) is None and g.pick(0.0, 0.0) == jf.JF_ERR_ARG
    assert jf.lib().jf_grid_rows(None) == jf.JF_ERR_ARG
    # KEMAR's description is recognised whatever memory it comes from, and is 710 rows
    k = jf.Grid.kemar()
    assert k.rows() == 710 and k.count.tolist() == KEMAR_COUNTS


def test_rings_from_measurement_directions():
    """jf_grid_from_positions: what a caller with a SOFA file's SourcePosition array does to get a jf_hrtf_grid and the row
    order of its impulse responses -- on KEMAR's own directions (whole degrees as the files are named: up to half a degree
    off the uniform ring), on a shuffled irregular set, and on sets that are not ring grids."""
    from jf_load import jf
    pos = model64.table_positions()                       # (elevation, rounded azimuth) of the 710 rows
    ele = np.array([e for e, _ in pos], np.float32)
    azi = np.array([a for _, a in pos], np.float32)
    g, row_of = jf.Grid.from_positions(azi, ele, tol_deg=0.51)
    assert g.count.tolist() == KEMAR_COUNTS and g.ele.tolist() == list(range(-40, 91, 10))
    assert np.array_equal(row_of, np.arange(710))         # the reference's loader order IS ring by ring, azimuth ascending
    assert g.rows() == 710
    # an irregular set in random order, azimuths given as 0 .. 360 with noise below the tolerance and one at 359.99
    e_r, c_r, _ = irregular_grid()
    rng = np.random.default_rng(4)
    el, az, want_row = [], [], []
    row = 0
    for r, n in enumerate(c_r):
        for i in range(n):
            el.append(min(90.0, e_r[r] + rng.uniform(-0.02, 0.02)))
            az.append((i * 360.0 / n + rng.uniform(-0.02, 0.02)) % 360.0)
            want_row.append(row)
            row += 1
    az[0] = 359.99
    perm = rng.permutation(len(el))
    g2, row_of2 = jf.Grid.from_positions(np.array(az)[perm], np.array(el)[perm], tol_deg=0.05)
    assert g2.count.tolist() == c_r and np.allclose(g2.ele, e_r, atol=0.03)
    assert np.array_equal(row_of2, np.array(want_row)[perm])
    assert np.allclose(g2.step[:-1], 360.0 / np.array(c_r[:-1], np.float32))
    for bad_az, bad_el in (([0, 90, 180, 275], [0, 0, 0, 0]),            # not uniform
                           ([10, 100, 190, 280], [0, 0, 0, 0]),           # does not start at azimuth 0
                           ([0, 180, 0, 180], [0, 0, 0.01, 0.01]),        # two rings closer than the tolerance: duplicates in one
                           ([0.0], [95.0])):
        with pytest.raises(jf.JfError) as ex:
            jf.Grid.from_positions(np.array(bad_az, np.float32), np.array(bad_el, np.float32), tol_deg=0.05)
        assert ex.value.code == jf.JF_ERR_ARG
