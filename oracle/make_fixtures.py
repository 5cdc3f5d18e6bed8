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


def main():
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
