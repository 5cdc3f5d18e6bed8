#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ (run in the build container).

DATA fixtures (copied measurements, not code), read from /root/reference:
  kemar_hrir_710x2x128_i16.npy   the 710x2 HRIR table the reference's loader
      (hrtf_signals.cu:107-153) would build, rebuilt from Jefferson/compact by
      mirroring (hrtf_signals.cpp:80-126 convention)
  kemar_positions_710x2_i16.npy  (elevation, round(azimuth)) of each row
  castanets_441_excerpt_i24.npy  first 2 s of Jefferson/media/Castanets-441.wav
      (the reference's default input, main.cu:16), raw 24-bit integers

MODEL fixtures (outputs of oracle/model64.py, float64 -- regression vectors of
this repo's own restatement, NOT outputs of the reference binary):
  golden_scenarios.npz           first blocks of the four benchmarkTesting
      scenarios (precision_test.cu:2154-2201) at B = 256 and B = 128
  interp_known.json              index/weight answers for every (ele, azi) the
      reference's tests and main.cu use
  xfade_reference_tests.npz      the reference's own stage-wise crossfade tests
      (xfadePrecisionTest precision_test.cu:455-1244: four old -> new pairs on the
      first block; xfadePrecisionCallbackTest :1248-2002: (8,18) -> (3,23) on three
      consecutive blocks) at B = 128 and 256: distance factor, weighted spectra of
      the old and of the new set, the crossfaded stereo block.  `--xfade` rebuilds
      this file alone from the committed data fixtures (no /root/reference needed).
"""
import json
import os
import sys
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import model64 as m  # noqa: E402

REF = "/root/reference/Jefferson"
OUT = os.path.join(HERE, "..", "tests", "golden")


def read_wav_int(path):
    with wave.open(path) as w:
        nch, sw, fs, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        a = np.frombuffer(raw, "<i2").astype(np.int32)
    elif sw == 3:
        b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
        a = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        a = np.where(a >= 1 << 23, a - (1 << 24), a).astype(np.int32)
    else:
        raise ValueError(sw)
    return a.reshape(n, nch), fs, sw


def build_hrir():
    pos = m.table_positions()
    assert len(pos) == m.NUM_HRTF
    tab = np.zeros((m.NUM_HRTF, 2, 128), np.int16)
    used = set()
    mirrored = 0
    for j, (e, a) in enumerate(pos):
        src_a = a if a <= 180 else 360 - a
        name = f"{REF}/compact/elev{e}/H{e}e{src_a:03d}a.wav"
        d, fs, sw = read_wav_int(name)
        assert fs == 44100 and sw == 2 and d.shape == (128, 2), (name, fs, sw, d.shape)
        used.add(name)
        if a <= 180:
            tab[j, 0], tab[j, 1] = d[:, 0], d[:, 1]
        else:  # left half-sphere: exchange L, R
            tab[j, 0], tab[j, 1] = d[:, 1], d[:, 0]
            mirrored += 1
    print(f"hrir rows {len(pos)}, mirrored {mirrored}, distinct files {len(used)}")
    return tab, np.array(pos, np.int16)


def scenario(hrir_f32, sig, B, azi0, ele0, n_dwell, n_rounds):
    """precision_test.cu:2093-2152 (CPU pass): dwell blocks, then azi += 5 per round."""
    mod = m.Model(B, 512, 1, hrir_f32)
    mod.set_signal(0, sig)
    mod.reset(0)
    mod.set_spherical(0, ele0, azi0, 0.5)
    out = []
    for _ in range(n_dwell):
        out.append(mod.process_block())
    azi = float(azi0)
    for _ in range(n_rounds):
        azi += 5
        if azi >= 360:
            azi -= 360
        mod.set_spherical(0, ele0, azi, 0.5)
        for _ in range(n_dwell):
            out.append(mod.process_block())
    return np.array(out)


# ---- the reference's crossfade tests (precision_test.cu:455-1244, 1248-2002) ------------------------------------------------
# old -> new as (ele, azi): precision_test.cu:505-508 (the source's fields), :727-728, :925-926, :1107-1108 (the arguments of
# interpolationCalculations(ele, azi)).  The tests set ele / azi only: the coordinates -- and with them the distance factor --
# stay the constructor's (0, 0, 0.5) (SoundSource.cu:8-13).  Their CPU crossfade is written the wrong way round
# (precision_test.cu:673): the kernel's / the production CPU path's formula is followed (kernels.cu:132-137).
XFADE_PAIRS = (((0, 0), (10, 5)), ((0, 3), (0, 8)), ((-5, 10), (5, 15)), ((8, 18), (3, 23)))
XFADE_CALLBACK_PAIR = ((8, 18), (3, 23))   # precision_test.cu:1298-1308: the same pair in each of the three rounds
XFADE_COORDS = (0.0, 0.0, 0.5)


def xfade_record(ele, azi):
    """the latched record {ele, azi, x, y, z} of a test position"""
    return np.array([ele, azi, *XFADE_COORDS], np.float32)


def xfade_stage_values(table, window, old, new, B):
    """What the reference's tests compare stage by stage for ONE window (float64): the distance factor D[513], the weighted
    spectra Y[set][ear][513] = sum_t w_t (X H[row_t][ear]) D of the old and of the new set (conv_bufs / intermediate,
    precision_test.cu:613-640), and the crossfaded last B frames [B][2] interleaved."""
    N = 1024
    X = np.fft.rfft(np.asarray(window, np.float64)) / N
    D = m.distance_factor(tuple(np.float32(c) for c in XFADE_COORDS), N // 2 + 1)
    Y = np.zeros((2, 2, N // 2 + 1), np.complex128)
    for k, (ele, azi) in enumerate((old, new)):
        h, om = m.interp(np.float32(ele), np.float32(azi))
        for row, w in m.terms(h, om):
            Y[k] += float(w) * (X[None, :] * table[row]) * D[None, :]
        Y[k, :, 0] = Y[k, :, 0].real
        Y[k, :, -1] = Y[k, :, -1].real     # c2r ignores Im of bins 0 and N/2
    y = np.fft.irfft(Y, n=N, axis=-1)[..., N - B:] * N
    fn = (np.arange(B, dtype=np.float32) / np.float32(B - 1.0)).astype(np.float32)   # kernels.cu:134
    out = y[0] * (np.float32(1.0) - fn).astype(np.float64)[None, :] + y[1] * fn.astype(np.float64)[None, :]
    return D, Y, out.T.copy()


def xfade_vectors(hrir_f32, sig):
    table = m.build_table(hrir_f32, 1024)
    gold = {}
    for B in (128, 256):
        def window(n_blocks):   # the window after n_blocks blocks of the input have been taken in
            w = np.zeros(1024, np.float32)
            n = min(n_blocks * B, 1024)
            w[1024 - n:] = sig[n_blocks * B - n: n_blocks * B]
            return w
        for p, (old, new) in enumerate(XFADE_PAIRS):
            D, Y, out = xfade_stage_values(table, window(1), old, new, B)
            gold[f"B{B}_pair{p}_dist"], gold[f"B{B}_pair{p}_Y"], gold[f"B{B}_pair{p}_out"] = D, Y, out.reshape(-1)
        for rnd in (1, 2, 3):
            D, Y, out = xfade_stage_values(table, window(rnd), *XFADE_CALLBACK_PAIR, B)
            gold[f"B{B}_cb{rnd}_dist"], gold[f"B{B}_cb{rnd}_Y"], gold[f"B{B}_cb{rnd}_out"] = D, Y, out.reshape(-1)
    return gold


def write_xfade(hrir_f32, sig):
    # the stage values are compared at the reference's 1e-6: stored as complex64 (4e-8 of the largest bin); the blocks, which
    # are compared at 2e-7, as float64
    gold = {k: (v.astype(np.complex64) if np.iscomplexobj(v) else v) for k, v in xfade_vectors(hrir_f32, sig).items()}
    np.savez_compressed(os.path.join(OUT, "xfade_reference_tests.npz"), **gold)


def main():
    if "--xfade" in sys.argv:   # from the committed data fixtures alone
        tab = np.load(os.path.join(OUT, "kemar_hrir_710x2x128_i16.npy"))
        ex = np.load(os.path.join(OUT, "castanets_441_excerpt_i24.npy"))
        write_xfade((tab.astype(np.float32) / np.float32(32768.0)).astype(np.float32),
                    (ex.astype(np.float64) / 8388608.0).astype(np.float32))
        print("wrote xfade_reference_tests.npz")
        return
    os.makedirs(OUT, exist_ok=True)
    tab, pos = build_hrir()
    np.save(os.path.join(OUT, "kemar_hrir_710x2x128_i16.npy"), tab)
    np.save(os.path.join(OUT, "kemar_positions_710x2_i16.npy"), pos)

    d, fs, sw = read_wav_int(f"{REF}/media/Castanets-441.wav")
    assert fs == 44100 and sw == 3 and d.shape[1] == 1
    ex = d[: 2 * 44100, 0].astype(np.int32)
    np.save(os.path.join(OUT, "castanets_441_excerpt_i24.npy"), ex)
    sig = (ex.astype(np.float64) / 8388608.0).astype(np.float32)  # libsndfile float scaling
    hrir = (tab.astype(np.float32) / np.float32(32768.0)).astype(np.float32)

    gold = {}
    # Short versions of the four benchmarkTesting scenarios: 3 dwell blocks per
    # position, 3 azimuth steps (reference: 172 dwell, 72 rounds).
    for B in (256, 128):
        for name, (azi0, ele0) in {"none": (0, 0), "azi": (3, 0), "ele": (0, 5), "both": (3, 5)}.items():
            gold[f"B{B}_{name}"] = scenario(hrir, sig, B, azi0, ele0, 3, 3)
    np.savez_compressed(os.path.join(OUT, "golden_scenarios.npz"), **gold)
    write_xfade(hrir, sig)

    known = {}
    pts = [(0, 0), (0, 3), (5, 0), (5, 3), (10, 5), (0, 8), (5, 15), (-5, 10), (3, 23), (8, 18),
           (4, 2), (3, 1), (2, 4), (9, 7), (0, 358), (-15, 7), (45, 10), (85, 20), (90, 0),
           (-40, 0), (-40, 359), (80, 345), (0, 360), (0, 355)]
    for ele, azi in pts:
        idx, om = m.interp(ele, azi)
        known[f"{ele},{azi}"] = {"idx": idx, "omegas": [float(o) for o in om],
                                 "case": m.case_of(idx)}
    with open(os.path.join(OUT, "interp_known.json"), "w") as f:
        json.dump({"azimuth_offset": m.AZIMUTH_OFFSET, "points": known}, f, indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
