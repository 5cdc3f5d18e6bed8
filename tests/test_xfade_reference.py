"""The reference's own stage-wise crossfade tests, by name.

`xfadePrecisionTest` (precision_test.cu:455-1244) takes the FIRST block of the default input (window = zeros + the first B
samples), sets old and new (ele, azi) by hand -- (0,0) -> (10,5) [:505-508], (0,3) -> (0,8) [:727-728], (-5,10) -> (5,15)
[:925-926], (8,18) -> (3,23) [:1107-1108] -- and compares its two paths stage by stage at 1e-6: distance factor, the weighted
spectra of the old and of the new filter set, both inverse transforms, the crossfade, the B stereo frames handed out.
`xfadePrecisionCallbackTest` (:1248-2002) does the same for (8,18) -> (3,23) [:1298-1308] on three consecutive blocks (rounds
1-3: `count` 0, B, 2 B; overlap-save in between).  The tests set ele / azi only, so the coordinates -- and the distance factor --
stay the constructor's (0, 0, 0.5) (SoundSource.cu:8-13); their CPU crossfade is written the wrong way round (:673, SURVEY.md
App. C#13): the kernel's formula is followed (kernels.cu:132-137).

Here: the float64 restatement of those stages is committed as tests/golden/xfade_reference_tests.npz (oracle/make_fixtures.py
--xfade).  CPU tests (this file, `not gpu`): the generator reproduces the file; the float32 C oracle and the float64 model, driven
through sessions that put exactly that window and that old -> new pair in front of them, reproduce its blocks.  GPU tests:
tests/test_gpu_xfade_reference.py.
"""
import os

import numpy as np
import pytest

import make_fixtures as mf
import model64
import oracle_lib
from conftest import GOLD, assert_within

TOL64, TOL32 = 2e-7, 4e-7   # the reference's end-to-end bound (precision_test.cu:2158) / two float32 paths


@pytest.fixture(scope="module")
def xgold():
    return np.load(os.path.join(GOLD, "xfade_reference_tests.npz"))


def cases(B):
    """(name in the fixture, old, new, blocks of input in the window)"""
    out = [(f"B{B}_pair{p}", old, new, 1) for p, (old, new) in enumerate(mf.XFADE_PAIRS)]
    out += [(f"B{B}_cb{r}", *mf.XFADE_CALLBACK_PAIR, r) for r in (1, 2, 3)]
    return out


def session_block(make_engine, sig, B, old, new, n_blocks, batch):
    """The block the reference's test forms: the window holds the first n_blocks blocks of the input, the source's old position
    is `old`, the position latched for the block is `new`.  Through the public calls only: a SILENT block at `old` first (the
    window stays zeros, old_ele / old_azi become `old`), then the signal (count = 0), n_blocks - 1 blocks at `old` and the
    block at `new`; batch: those n_blocks as one batch call."""
    e = make_engine()
    e.set_signal(0, np.zeros(0, np.float32))
    rec = lambda p: np.tile(mf.xfade_record(*p), (1, 1, 1))   # [1][1][5]
    e.process_batch(rec(old))
    e.set_signal(0, sig)
    pos = np.concatenate([rec(old)] * (n_blocks - 1) + [rec(new)], axis=0)
    if batch:
        out = e.process_batch(pos)
        out = out[0] if isinstance(out, tuple) else out
        blk = np.asarray(out)[-1]
    else:
        for k in range(n_blocks):
            out = e.process_batch(pos[k:k + 1])
            out = out[0] if isinstance(out, tuple) else out
            blk = np.asarray(out)[-1]
    if hasattr(e, "close"):
        e.close()
    return blk


def test_fixture_is_what_the_generator_makes(hrir, castanets, xgold):
    fresh = mf.xfade_vectors(hrir, castanets)
    assert sorted(fresh) == sorted(xgold.files)
    for k in fresh:
        want = fresh[k]
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(xgold[k] - want).max() <= (1e-7 if np.iscomplexobj(want) else 1e-15) * scale, k


def test_known_answers_of_the_pairs():
    """the index / weight answers the reference's tests rest on: cases 1, 2, 3 / 1 (the negative-elevation
    quirk), 4"""
    want_case = {(0, 0): 1, (10, 5): 1, (0, 3): 2, (0, 8): 2, (-5, 10): 1, (5, 15): 3, (8, 18): 4, (3, 23): 4}
    for (ele, azi), c in want_case.items():
        h, om = model64.interp(np.float32(ele), np.float32(azi))
        assert model64.case_of(h) == c, (ele, azi)


@pytest.mark.parametrize("B", [128, 256])
def test_model_and_oracle_sessions_reproduce_the_reference_tests(hrir, castanets, xgold, B):
    for name, old, new, nb in cases(B):
        want = xgold[name + "_out"]
        m = session_block(lambda: model64.Model(B, 512, 1, hrir), castanets, B, old, new, nb, batch=True)
        assert np.abs(m - want).max() <= 1e-12, name          # the model IS the restatement: same float64 steps
        for batch in (False, True):
            o = session_block(lambda: oracle_lib.Engine(B, 512, 1, hrir), castanets, B, old, new, nb, batch)
            assert_within(o, want, TOL64 * 1.5, f"xfade {name} oracle32 batch={batch}")   # one float32 path against float64
