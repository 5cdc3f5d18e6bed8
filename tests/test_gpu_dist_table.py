"""The distance-factor tables (include/jefferson.h: jf_debug_set_distance_table): the 513 factors of a block
(generateDistanceFactor, kernels.cu:116-125) depend on |coords| alone, which takes few distinct float32 values per source over
a trajectory; the upload evaluates a table for up to four of them per source and the batch kernels load the factors instead of
evaluating them per block.  The claim is bit-identity with the per-block evaluation -- a table holds what the same device
function yields -- whichever items read tables."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu

TOL32 = 4e-7


def _trajectory(jf, S, K, seed=0):
    rng = np.random.default_rng(seed)
    pos = np.zeros((K, S, 5), np.float32)
    r0 = rng.uniform(0.3, 4.0, S)
    for s in range(S):
        for k in range(K):
            r = r0[s]
            if 12 <= s < 16 and k >= 5 + s:      # these sources change their distance once in the run: two or three values more
                r = r0[s] * 1.25
            if 16 <= s < 18:                     # ... and these in every block: more values than tables
                r = r0[s] * (1.0 + 0.01 * k)
            azi = (17 * s + (0 if s == 19 else k)) % 360   # source 19 does not move at all: one value
            pos[k, s] = jf.position_from_spherical(-40 + (11 * s) % 131, azi, r)
    if S > 18:
        pos[:, 18, 2] = np.nan                    # a source with unusable coordinates: silent, and no table
    return pos


def _run(jf, hrir, pos, on, windows, mode=None, B=256, group=4):
    K, S = pos.shape[0], pos.shape[1]
    e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=max(k for _, k in windows))
    rng = np.random.default_rng(3)
    sigs = [rng.uniform(-0.5, 0.5, 2500 + 31 * s).astype(np.float32) for s in range(S)]
    for s in range(S):
        e.set_signal(s, sigs[s])
    e.set_source_group(group)
    e.set_distance_table(on)
    if mode is not None:
        e.set_mode(mode)
    e.upload_positions(pos)
    n_tab = e.distance_table_share()
    out = []
    for first, k in windows:
        e.batch_run(first, k)
        e.synchronize()
        out.append(e.read_device(e.partial_device_ptr(), (k, S // max(group, 1), 2 * B)))
    e.close()
    return np.concatenate(out), n_tab, sigs


def test_tables_where_the_distance_takes_few_values_and_the_same_bits(jf, hrir):
    S, K = 24, 40
    pos = _trajectory(jf, S, K)
    windows = [(0, 16), (16, 16), (32, 8)]
    a, n_on, sigs = _run(jf, hrir, pos, True, windows)
    b, n_off, _ = _run(jf, hrir, pos, False, windows)
    assert n_off == 0
    # sources that move in azimuth only (|coords| flickers by an ulp: two or three values) or change their distance once
    # read tables throughout; the two whose distance changes every block have tables for their first four values only; the
    # one with NaN coordinates has none
    assert 800 <= n_on < 1000 * (S - 1) // S, n_on
    assert np.abs(a).max() > 0.02
    assert np.array_equal(a, b)
    # per-source kernel (single sources per unit) and B = 128 read the tables too
    for B, group in ((256, 1), (128, 2)):
        a1, _, _ = _run(jf, hrir, pos, True, windows, B=B, group=group)
        b1, _, _ = _run(jf, hrir, pos, False, windows, B=B, group=group)
        assert np.array_equal(a1, b1)
    # and against the oracle (the NaN source is silent on both sides)
    ora = oracle_lib.Engine(256, 512, S, hrir)
    for s in range(S):
        ora.set_signal(s, sigs[s])
    _, part = ora.process_batch(pos, want_partial=True)            # [S][K][2B]
    want = np.nan_to_num(part.astype(np.float64)).reshape(S // 4, 4, K, 512).sum(axis=1).transpose(1, 0, 2)
    assert np.abs(a - want).max() <= TOL32 * 4 * max(1.0, np.abs(want).max())


def test_tables_follow_the_trajectory_and_the_mode(jf, hrir):
    """A second upload with other distances rebuilds the tables; FD_BASIC (no distance factor at all) never reads them;
    per-block calls and positions handed over outside a trajectory evaluate as before."""
    S, K = 8, 12
    e_on = e_off = None
    outs = {}
    for on in (True, False):
        e = jf.Engine(256, 512, S, hrir=hrir, max_batch_blocks=K)
        rng = np.random.default_rng(4)
        for s in range(S):
            e.set_signal(s, rng.uniform(-0.5, 0.5, 3000).astype(np.float32))
        e.set_source_group(4)
        e.set_distance_table(on)
        got = []
        for seed in (1, 2):
            pos = _trajectory(jf, S, K, seed=seed)      # sources 0..7: all keep their distance within a trajectory
            e.upload_positions(pos)
            assert e.distance_table_share() == (1000 if on else 0)
            e.batch_run(0, K)
            e.synchronize()
            got.append(e.read_device(e.mix_device_ptr(), (K, 512)))
        e.set_mode(jf.JF_MODE_FD_BASIC)
        e.batch_run(0, K)
        e.synchronize()
        got.append(e.read_device(e.mix_device_ptr(), (K, 512)))
        e.set_mode(jf.JF_MODE_FD_COMPLEX)
        for k in range(3):                               # per-block calls: the real-time kernel, no trajectory
            for s in range(S):
                e.set_spherical(s, 10, (40 * s + 5 * k) % 360, 0.5 + 0.3 * s)
            got.append(e.process_block()[None, :])
        outs[on] = np.concatenate(got)
        e.close()
    assert np.abs(outs[True]).max() > 0.02
    assert np.array_equal(outs[True], outs[False])
    assert not np.array_equal(outs[True][:K], outs[True][K:2 * K])     # the second trajectory really differs
