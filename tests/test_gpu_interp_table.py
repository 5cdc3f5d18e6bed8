"""The pre-interpolated rows of the HRTF table (jf_device.h: htab rows 710 ..; include/jefferson.h:
JF_FLAG_NO_INTERP_TABLE): for every whole-degree position the setters can latch (SoundSource.cu:33-34,42-43) the weighted
filter sum_t w_t H[row_t] that GPUSoundSource.cu:118-292 forms per block, built once.  The claim is bit-identity with the
per-block weighting; it is tested end to end over ALL 131 x 360 positions, the rows themselves against the index/weight
rule, and the mixed cases (fractional positions, positions outside the rows' range, the corrected rule, FD_BASIC)."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu

TOL32 = 4e-7


def _engine(jf, hrir, S, K, B=256, flags=0, group=2, sig_seed=5):
    e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=K, flags=flags)
    rng = np.random.default_rng(sig_seed)
    sigs = [rng.uniform(-0.5, 0.5, 3000 + 17 * s).astype(np.float32) for s in range(S)]
    for s in range(S):
        e.set_signal(s, sigs[s])
    if group:
        e.set_source_group(group)
    return e, sigs


def test_every_whole_degree_position_bit_identical_to_the_per_block_weighting(jf, hrir):
    """262 sources x 360 blocks: source s sits at elevation -40 + s mod 131 and sweeps the 360 azimuths one degree per
    block (two sources per elevation, half a circle apart), so every one of the 47 160 rows is the new set of two
    source-blocks and the old set of two more.  One engine reads the pre-interpolated rows, the other weights the
    measured rows per block: every float of every unit's stereo block must be the same."""
    S, K, B = 262, 360, 256
    ele = np.array([-40 + s % 131 for s in range(S)], np.float32)
    azi0 = np.array([180 * (s // 131) for s in range(S)], np.int64)
    r = np.linspace(0.3, 3.0, S).astype(np.float32)
    b = np.arange(K, dtype=np.int64)[:, None]
    pos = jf.positions_from_spherical(np.broadcast_to(ele, (K, S)), ((azi0[None, :] + b) % 360).astype(np.float32),
                                      np.broadcast_to(r, (K, S)))
    out = []
    for on in (True, False):
        e, _ = _engine(jf, hrir, S, K)
        assert e.interp_table() == 2        # default: decided per run
        e.set_interp_table(on)              # here: always / never
        assert e.interp_table() == int(on)
        e.upload_positions(pos)
        e.batch_run(0, K)
        e.synchronize()
        assert any(k.startswith("fused_pair_kernel") for k in e.last_kernels())
        # every item but the first block's (which fade in from (0, 0): also a row, ele 0 azi 0) is a crossfade of two rows
        n_pre = e.count_desc_flags(K * S, 4)
        assert n_pre == (K * S if on else 0), n_pre
        out.append((e.read_device(e.partial_device_ptr(), (K, S // 2, 2 * B)), e.read_device(e.mix_device_ptr(), (K, 2 * B))))
        e.close()
    assert np.abs(out[0][0]).max() > 0.05
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])


def test_rows_are_the_weighted_sums_of_the_rule(jf, hrir):
    """Row 710 + (ele + 40) 360 + azi against sum_t w_t H[row_t] with rows and weights from the index/weight kernel and
    H from the measured part of the table, evaluated in float64: within float32 rounding of a four-term sum, for all
    47 160 rows -- i.e. every row belongs to ITS position (the bit-level claim is the test above)."""
    e = jf.Engine(256, 512, 1, hrir=hrir)
    base = e.read_table_rows(0, jf.NUM_HRTF).astype(np.float64)          # [710][512][4]
    ee, aa = np.meshgrid(np.arange(-40, 91), np.arange(360), indexing="ij")
    rows, w, nt = e.interp_device(ee.ravel().astype(np.float32), aa.ravel().astype(np.float32))
    assert (nt > 0).all()        # no whole-degree position inside the range is uninterpolable
    worst = 0.0
    CH = 131 * 360 // 10
    for c in range(10):
        got = e.read_table_rows(jf.NUM_HRTF + c * CH, CH).astype(np.float64)
        rr, ww, nn = rows[c * CH:(c + 1) * CH], w[c * CH:(c + 1) * CH].astype(np.float64), nt[c * CH:(c + 1) * CH]
        want = np.zeros_like(got)
        mag = np.zeros(got.shape[:1])
        for t in range(4):
            use = (t < nn).astype(np.float64)
            want += (use * ww[:, t])[:, None, None] * base[rr[:, t]]
            mag += use * np.abs(ww[:, t]) * np.abs(base[rr[:, t]]).max(axis=(1, 2))
        err = np.abs(got - want).max(axis=(1, 2)) / mag
        worst = max(worst, float(err.max()))
    e.close()
    assert worst <= 2.5e-7, worst


def test_mixed_positions_and_the_oracle(jf, hrir):
    """Sources on whole degrees beside sources on fractional ones, below -40 degrees (outside the rows: the reference's
    truncating rule extrapolates there) and a source that alternates between the two kinds from block to block (one
    pre-interpolated set beside an ordinary one): bit-identical to an engine without the rows, and both within the
    float32 tolerance of the C oracle."""
    S, K, B = 8, 24, 128
    pos = np.zeros((K, S, 5), np.float32)
    for k in range(K):
        pos[k, 0] = jf.position_from_spherical(5, (3 * k) % 360, 1.0)            # rows
        pos[k, 1] = jf.position_from_spherical(5.5, (3 * k) % 360 + 0.25, 1.0)   # fractional: never rows
        pos[k, 2] = jf.position_from_spherical(-45, (7 * k) % 360, 0.7)          # whole degrees but below the rows' range
        pos[k, 3] = jf.position_from_spherical(20, 10 + (0.5 if k % 2 else 0.0) * 1, 2.0)   # alternates row / weighted
        pos[k, 4] = jf.position_from_spherical(90, (11 * k) % 360, 1.5)          # the pole ring
        pos[k, 5] = jf.position_from_spherical(-40, 359 - (k % 3), 0.4)          # lowest ring, last azimuths (no wrap)
        pos[k, 6] = jf.position_from_spherical(37, 123, 1.0)                      # does not move: its new set is its old set
        pos[k, 7] = jf.position_from_spherical(0, 0, 0.5) if k < 12 else jf.position_from_spherical(0, 1, 0.5)
    outs = []
    for on in (True, False):
        e, sigs = _engine(jf, hrir, S, K, B=B, group=4)
        e.set_interp_table(on)
        e.upload_positions(pos)
        e.batch_run(0, K)
        e.synchronize()
        if on:
            n4 = e.count_desc_flags(K * S, 4)
            assert 0 < n4 < K * S       # some items take the rows, some do not
        outs.append(e.read_device(e.partial_device_ptr(), (K, S // 4, 2 * B)))
        e.close()
    assert np.array_equal(outs[0], outs[1])
    ora = oracle_lib.Engine(B, 512, S, hrir)
    for s in range(S):
        ora.set_signal(s, sigs[s])
    _, part = ora.process_batch(pos, want_partial=True)          # [S][K][2B]
    want = part.astype(np.float64).reshape(2, 4, K, 2 * B).sum(axis=1).transpose(1, 0, 2)
    assert np.abs(want).max() > 0.05
    assert np.abs(outs[0] - want).max() <= TOL32 * 4 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("what", ["corrected", "basic", "no_table"])
def test_other_rules_and_modes(jf, hrir, what):
    """The corrected index/weight rule gets rows built with THAT rule; FD_BASIC (nearest measured row) never reads them; an
    engine created without them (JF_FLAG_NO_INTERP_TABLE) refuses to switch them on and works as before."""
    S, K, B = 16, 40, 256
    ids = np.arange(S)
    ele = (-40 + (ids * 9) % 131).astype(np.float32)
    b = np.arange(K, dtype=np.int64)[:, None]
    azi = ((ids * 23)[None, :] + 2 * b) % 360
    pos = jf.positions_from_spherical(np.broadcast_to(ele, (K, S)), azi.astype(np.float32),
                                      np.broadcast_to(np.float32(1.2), (K, S)))
    flags = jf.JF_FLAG_CORRECTED_INTERPOLATION if what == "corrected" else 0
    outs = []
    for on in (True, False):
        f = flags | (jf.JF_FLAG_NO_INTERP_TABLE if (what == "no_table" and on) else 0)
        e, sigs = _engine(jf, hrir, S, K, flags=f, group=4)
        if what == "no_table" and on:
            assert not e.interp_table()
            with pytest.raises(jf.JfError) as ex:
                e.set_interp_table(True)
            assert ex.value.code == jf.JF_ERR_STATE
        else:
            e.set_interp_table(on)
        if what == "basic":
            e.set_mode(jf.JF_MODE_FD_BASIC)
        e.upload_positions(pos)
        e.batch_run(0, K)
        e.synchronize()
        n4 = e.count_desc_flags(K * S, 4)
        assert n4 == (K * S if (on and what == "corrected") else 0), n4
        outs.append(e.read_device(e.partial_device_ptr(), (K, S // 4, 2 * B)))
        e.close()
    assert np.abs(outs[0]).max() > 0.01
    assert np.array_equal(outs[0], outs[1])
    if what == "corrected":
        ora = oracle_lib.Engine(B, 512, S, hrir)
        ora.set_mode(2)       # the corrected rule
        for s in range(S):
            ora.set_signal(s, sigs[s])
        _, part = ora.process_batch(pos, want_partial=True)
        want = part.astype(np.float64).reshape(S // 4, 4, K, 2 * B).sum(axis=1).transpose(1, 0, 2)
        assert np.abs(outs[0] - want).max() <= TOL32 * 4 * max(1.0, np.abs(want).max())


def test_the_rows_are_taken_per_run_by_how_many_items_move(jf, hrir):
    """The default setting decides per run of an uploaded trajectory: sources that stay read one cached row per block,
    sources that move every block would stream a new row from HBM per block, which is not faster than the weighting
    (profiles/r04/interp_table.md) -- so a window takes the rows unless more than 30 % of its items move (the measured crossover is a third).
    Whatever is chosen, the blocks are the same bit for bit as with the rows always or never."""
    S, K, B = 16, 16, 128
    ids = np.arange(S)
    ele = (-40 + (ids * 9) % 131).astype(np.float32)

    def traj(move_every):
        b = np.arange(4 * K, dtype=np.int64)[:, None]
        azi = ((ids * 23)[None, :] + b // move_every) % 360
        return jf.positions_from_spherical(np.broadcast_to(ele, azi.shape), azi.astype(np.float32),
                                           np.broadcast_to(np.float32(0.9), azi.shape))
    for move_every, expect in ((1, False), (2, False), (4, True), (172, True)):
        pos = traj(move_every)
        outs = {}
        for setting in (2, 1, 0):
            e, _ = _engine(jf, hrir, S, K, B=B, group=4)
            e.set_interp_table(setting)
            e.upload_positions(pos)
            blocks = []
            for w in range(4):
                e.batch_run(w * K, K)
                e.synchronize()
                if setting == 2:
                    assert e.last_run_used_rows() == expect, (move_every, w)
                else:
                    assert e.last_run_used_rows() == bool(setting)
                blocks.append(e.read_device(e.mix_device_ptr(), (K, 2 * B)))
            outs[setting] = np.concatenate(blocks)
            e.close()
        assert np.abs(outs[2]).max() > 0.01
        assert np.array_equal(outs[2], outs[1]) and np.array_equal(outs[2], outs[0])


def test_the_rows_are_built_by_the_first_run_that_takes_them(jf, hrir):
    """386 MB of rows nobody reads are not built: an engine holds the 710 measured rows until a run's policy takes the
    pre-interpolated ones (or a caller asks for them), whatever runs before -- per-block calls, runs in which every source
    moves -- and the run that builds them gives the same blocks, bit for bit, as an engine that never has them.  An engine
    that may not have them (JF_FLAG_NO_INTERP_TABLE) never builds them."""
    S, K, B = 16, 16, 128
    ids = np.arange(S)
    ele = (-40 + (ids * 9) % 131).astype(np.float32)
    b = np.arange(3 * K, dtype=np.int64)[:, None]
    moving = ((ids * 23)[None, :] + b) % 360
    azi = np.where(b < K, moving, ((ids * 23)[None, :] + K - 1) % 360)      # window 0 moves every block, windows 1 and 2 stay
    pos = jf.positions_from_spherical(np.broadcast_to(ele, azi.shape), azi.astype(np.float32),
                                      np.broadcast_to(np.float32(0.9), azi.shape))
    e, _ = _engine(jf, hrir, S, K, B=B, group=4)
    never, _ = _engine(jf, hrir, S, K, B=B, group=4, flags=jf.JF_FLAG_NO_INTERP_TABLE)
    assert e.interp_table() == 2 and not e.interp_table_built()
    assert never.interp_table() == 0 and not never.interp_table_built()
    for x in (e, never):
        x.process_block()                    # the one-launch kernel: no rows
        x.upload_positions(pos)
    assert not e.interp_table_built()
    for w in range(3):
        for x in (e, never):
            x.batch_run(w * K, K)
            x.synchronize()
        assert e.last_run_used_rows() == (w > 0) and not never.last_run_used_rows()
        assert e.interp_table_built() == (w > 0), w      # built by window 1, the first one whose policy takes them
        assert not never.interp_table_built()
        got = e.read_device(e.mix_device_ptr(), (K, 2 * B))
        want = never.read_device(never.mix_device_ptr(), (K, 2 * B))
        assert np.abs(want).max() > 0.01 and np.array_equal(got, want), w
    # asking for them builds them at once
    e2, _ = _engine(jf, hrir, S, K, B=B, group=4)
    assert not e2.interp_table_built()
    e2.set_interp_table(1)
    assert e2.interp_table_built()
    rows = e2.read_table_rows(jf.NUM_HRTF, 4)
    assert np.array_equal(rows, e.read_table_rows(jf.NUM_HRTF, 4)) and np.abs(rows).max() > 0
    for x in (e, e2, never):
        x.close()
