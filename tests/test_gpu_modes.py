"""Data::type mode switch: JF_MODE_FD_BASIC = the reference's nearest-HRTF path
(CPUSoundSource.cpp:113-142), which is also what its time-domain modes compute."""
import numpy as np
import pytest

import model64
import oracle_lib
from conftest import assert_within, sum_tol

pytestmark = pytest.mark.gpu

TOL64 = 2e-7
TOL32 = 4e-7


@pytest.mark.parametrize("S,rt_max", [(1, 16), (3, 0), (20, 16)])
def test_fd_basic_vs_oracles_and_time_domain(jf, hrir, castanets, S, rt_max):
    """Per-block calls through both the one-launch kernel and the batch pipeline."""
    B, K = 256, 8
    eng = jf.Engine(B, 512, S, hrir=hrir)
    eng.set_rt_max_sources(rt_max)
    eng.set_mode(jf.JF_MODE_FD_BASIC)
    ora = oracle_lib.Engine(B, 512, S, hrir)
    ora.set_mode(1)
    mod = model64.Model(B, 512, S, hrir)
    mod.mode = 1
    sigs = [0.5 * castanets[3000 + 500 * s: 3000 + 500 * s + K * B] for s in range(S)]
    for s in range(S):
        for x in (eng, ora, mod):
            x.set_signal(s, sigs[s])
    got, w32, w64 = [], [], []
    pos_of = {}
    for k in range(K):
        for s in range(S):
            ele, azi = -38 + (7 * s + 3 * k) % 125, (31 * s + 9 * k) % 360   # moves every block: no crossfade in this mode
            pos_of[(k, s)] = (ele, azi)
            for x in (eng, ora, mod):
                x.set_spherical(s, ele, azi, 0.5 + 0.1 * s)
        got.append(eng.process_block())
        w32.append(ora.process_block())
        w64.append(mod.process_block())
    eng.close()
    got, w32, w64 = np.array(got), np.array(w32), np.array(w64)
    assert np.abs(w64).max() > 0.02
    assert_within(got, w64, sum_tol(TOL64, S), f'FD_BASIC S={S} vs model64')
    assert_within(got, w32, sum_tol(TOL32, S), f'FD_BASIC S={S} vs oracle32')
    if S == 1:
        # time-domain definition (CPU_TD): y[n] = sum_k x[n-k] h[k] with the nearest HRIR of each block
        stream = sigs[0].astype(np.float64)
        for k in range(K):
            idx = model64.pick_hrtf(*pos_of[(k, 0)])
            for ear in range(2):
                y = np.convolve(stream, hrir[idx, ear].astype(np.float64))[k * B:(k + 1) * B]
                assert np.abs(got[k, ear::2] - y).max() <= TOL64


def test_mode_switch_between_blocks(jf, hrir, castanets):
    """Data::type is read at every block (Audio.cu:104): switching modes mid-stream keeps the window
    and the play position; batch calls honour it too."""
    B = 128
    eng = jf.Engine(B, 512, 2, hrir=hrir, max_batch_blocks=4)
    ora = oracle_lib.Engine(B, 512, 2, hrir)
    for x in (eng, ora):
        for s in range(2):
            x.set_signal(s, castanets[1000 * s: 1000 * s + 6000])
            x.set_spherical(s, 12, 45 + 100 * s, 0.7)
    for k in range(10):
        mode = (k // 3) % 2
        eng.set_mode(mode)
        ora.set_mode(mode)
        if k == 5:
            for x in (eng, ora):
                x.set_spherical(0, 12, 50, 0.7)
        assert np.abs(eng.process_block() - ora.process_block()).max() <= TOL32
    pos = np.tile(np.stack([jf.position_from_spherical(12, 50, 0.7), jf.position_from_spherical(12, 145, 0.7)]), (4, 1, 1))
    eng.set_mode(jf.JF_MODE_FD_BASIC)
    ora.set_mode(1)
    assert np.abs(eng.process_batch(pos) - ora.process_batch(pos)).max() <= TOL32
    with pytest.raises(jf.JfError):
        eng.set_mode(7)
    eng.close()
