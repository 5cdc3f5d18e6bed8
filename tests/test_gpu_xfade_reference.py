"""GPU: the reference's own stage-wise crossfade tests, by name, through the HIP path (the reading of
precision_test.cu:455-1244 and :1248-2002 is in tests/test_xfade_reference.py).  Per case -- xfadePrecisionTest's four old -> new
pairs on the first block, xfadePrecisionCallbackTest's (8,18) -> (3,23) on blocks 1, 2, 3 -- and for B = 128 and 256:
  * the distance factor and the weighted spectra of the OLD and of the NEW filter set, through the device code of the fused
    kernels (jf_debug_stage_taps), at the reference's 1e-6 (of the largest bin) against the committed float64 vectors;
  * the B crossfaded stereo frames the engine hands out, per-block (jf_process_block: the one-launch kernel) and batch
    (jf_process_batch: prep -> fused -> mix), at 2e-7 against the float64 vectors and 4e-7 against the C oracle in the same
    session."""
import os

import numpy as np
import pytest

import make_fixtures as mf
import oracle_lib
from conftest import GOLD, assert_within
from test_xfade_reference import TOL32, TOL64, cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def xgold():
    return np.load(os.path.join(GOLD, "xfade_reference_tests.npz"))


def window(sig, B, n_blocks):
    w = np.zeros(1024, np.float32)
    n = min(n_blocks * B, 1024)
    w[1024 - n:] = sig[n_blocks * B - n: n_blocks * B]
    return w


@pytest.mark.parametrize("B", [128, 256])
def test_xfade_precision_test_stage_values(jf, hrir, castanets, xgold, B):
    """distance factor + the weighted spectra of both sets (`Inaccurate Distance calculations` / `Inaccurate Case n
    Convolutions` of the reference's test) at its own 1e-6"""
    e = jf.Engine(B, 512, 1, hrir=hrir)
    for name, old, new, nb in cases(B):
        pos = np.stack([mf.xfade_record(*old), mf.xfade_record(*new)])
        wins = np.stack([window(castanets, B, nb)] * 2)
        D, Y = e.stage_taps(pos, wins)
        wantD, wantY = xgold[name + "_dist"], xgold[name + "_Y"]
        for k in range(2):
            assert np.abs(D[k, :512] - wantD[:512]).max() <= 1e-6, (name, k)          # |D| < 1: absolute, as the reference's
            assert abs(D[k, 512].real - wantD[512].real) <= 1e-6, (name, k)            # (c2r never reads Im of bin 512)
            scale = float(np.abs(wantY[k]).max())
            assert scale > 1e-6, name
            assert np.abs(Y[k] - wantY[k]).max() <= 1e-6 * scale, (name, "old" if k == 0 else "new")
    e.close()


@pytest.mark.parametrize("B", [128, 256])
@pytest.mark.parametrize("batch", [False, True], ids=["per_block", "batch"])
def test_xfade_precision_test_blocks(jf, hrir, castanets, xgold, B, batch):
    """the crossfaded B stereo frames (`Successfully accurate case n output`): the window holds the first n blocks of the
    default input, the source's old position is the test's old pair, the block is latched at its new pair"""
    for name, old, new, nb in cases(B):
        eng = jf.Engine(B, 512, 1, hrir=hrir, max_batch_blocks=4)
        ora = oracle_lib.Engine(B, 512, 1, hrir)
        got = {}
        for x in (eng, ora):
            rec = lambda p: np.tile(mf.xfade_record(*p), (1, 1, 1))
            per_block = not batch and x is eng     # (the oracle's batch call IS its per-block loop)
            x.set_signal(0, np.zeros(0, np.float32))   # a silent block at `old`: the window stays zeros, old := `old`
            if per_block:
                x.set_latched(rec(old)[0])
                x.process_block()
            else:
                x.process_batch(rec(old))
            x.set_signal(0, castanets)
            pos = np.concatenate([rec(old)] * (nb - 1) + [rec(new)], axis=0)
            if per_block:
                for k in range(nb):
                    x.set_latched(pos[k])
                    got[x] = np.asarray(x.process_block())
            else:
                got[x] = np.asarray(x.process_batch(pos))[-1]
        want = xgold[name + "_out"]
        assert float(np.abs(want).max()) > 1e-4, name
        assert_within(got[eng], want, TOL64, f"xfade {name} hip vs float64 batch={batch}")
        assert_within(got[eng], got[ora], TOL32, f"xfade {name} hip vs oracle32 batch={batch}")
        eng.close()
        ora.close()
