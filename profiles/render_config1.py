"""BASELINE.json configs[0]/[1] through the plain-C offline driver: 1 source, 256-sample blocks, compact KEMAR
set, WAV in -> WAV out, the complete benchmarkTesting trajectory (172 blocks x 73 positions = 12 556 blocks),
next to the float32 C oracle (the CPU overlap-save reference restated) timed on one host core."""
import os
import struct
import subprocess
import sys
import tempfile
import time
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import model64  # noqa: E402
import oracle_lib  # noqa: E402

hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
ex = np.load(os.path.join(ROOT, "tests/golden/castanets_441_excerpt_i24.npy"))
tmp = tempfile.mkdtemp()
kemar = os.path.join(tmp, "compact")
for j, (e, a) in enumerate(model64.table_positions()):
    if a > 180:
        continue
    d = os.path.join(kemar, f"elev{e}")
    os.makedirs(d, exist_ok=True)
    with wave.open(os.path.join(d, f"H{e}e{a:03d}a.wav"), "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(44100)
        w.writeframes(np.round(hrir[j].T * 32768.0).astype(np.int16).tobytes())
inp, outp = os.path.join(tmp, "in.wav"), os.path.join(tmp, "out.wav")
with wave.open(inp, "wb") as w:
    w.setnchannels(1)
    w.setsampwidth(3)
    w.setframerate(44100)
    w.writeframes(b"".join(struct.pack("<i", int(v))[:3] for v in ex))
outs = {}
for extra in ([], ["--latency"], ["--batch", "64"], ["--batch", "512"]):
    r = subprocess.run([os.path.join(ROOT, "jefferson-2.0_amd", "jf_render"), kemar, inp, outp] + extra,
                       capture_output=True, text=True)
    print("jf_render", " ".join(extra), "->", r.stderr.strip().splitlines()[-1])
    with wave.open(outp, "rb") as w:
        outs[" ".join(extra)] = w.readframes(w.getnframes())
# the batch render is the same audio as the per-block render (24-bit PCM: identical files)
print("batch render identical to per-block render:", outs["--batch 64"] == outs[""], outs["--batch 512"] == outs[""])

# the CPU restatement on the same job, one thread
sig = (ex / 8388608.0).astype(np.float32)
ora = oracle_lib.Engine(256, 512, 1, hrir)
ora.set_signal(0, sig)
ora.reset(0)
pos = []
azi = 3.0
for rnd in range(73):
    if rnd:
        azi = (azi + 5) % 360
    pos += [oracle_lib.from_spherical(5, azi, 0.5)] * 172
pos = np.stack(pos)[:, None, :]
t0 = time.perf_counter()
ora.process_batch(pos, n_threads=1)
dt = time.perf_counter() - t0
print(f"C oracle, 1 thread: {len(pos)} blocks in {dt:.3f} s: {1e6 * dt / len(pos):.1f} us per block, "
      f"real-time factor {len(pos) * 256 / 44100 / dt:.1f}")
