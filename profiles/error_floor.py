"""Round 4, VERDICT item 7: which stage owns the full-scale error floor?  Noise of amplitude 0.5 at r = 0.05 (the case of
tests/test_gpu_engine.py::test_distance_sweep_gain_and_delay that sits at 2.0-2.8e-7 worst sample / 4.2e-8 rms against float64)
for six seeds.  Stage outputs of the device code (jf_debug_rfft_device: forward transform; jf_debug_stage_taps: distance
factor and weighted spectra Y = sum_t w_t X H_t D) are carried on in FLOAT64 to the output, so that each stage's share of the
output error can be read off:
    e_fwd   : device forward transform, everything behind it in float64
    e_spec  : device spectra Y (forward + distance factor + filter), inverse in float64
    e_total : the device's output (jf_process_block)
    D alone, filter alone: the device's D / the device's weighted sum with float64 inputs otherwise (from the taps)
Errors are rms over the block's 2 x 256 output samples, 8 blocks per seed (behind the first, which fades in from (0, 0)), in units of 1e-8."""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
from jf_load import jf
import model64
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
B, N, NC = 256, 1024, 513
table = model64.build_table(hrir, N)
rows = []
for seed in (11, 12, 13, 14, 15, 16):
    rng = np.random.default_rng(seed)
    sig = rng.uniform(-.5, .5, 8192).astype(np.float32)
    e = jf.Engine(B, 512, 1, hrir=hrir)
    e.set_signal(0, sig)
    e.set_spherical(0, 0, 45, 0.05)
    ele, azi, coords = model64.from_spherical(0, 45, 0.05)
    pos = jf.position_from_spherical(0, 45, 0.05)
    cur = model64.interp(ele, azi)
    D64 = model64.distance_factor(coords, NC)
    x = np.zeros(N)
    acc = {k: 0.0 for k in ("fwd", "dist", "filt", "spec", "total")}
    worst = 0.0
    n = 0
    for blk in range(9):
        x[:N - B] = x[B:].copy()
        x[N - B:] = sig[(blk * B + np.arange(B)) % len(sig)].astype(np.float64)
        y_dev = e.process_block().astype(np.float64).reshape(B, 2).T          # [2][B]
        win = x.astype(np.float32)[None, :]
        X64 = np.fft.rfft(x) / N

        def out_of(Y):            # float64 inverse of spectra [2][513] -> the block's frames
            Y = Y.copy()
            Y[:, 0] = Y[:, 0].real
            Y[:, -1] = Y[:, -1].real
            return (np.fft.irfft(Y, n=N, axis=-1) * N)[:, N - B:]

        def spectra(X, D):        # float64 weighted filter, GPU order of operations does not matter here
            Y = np.zeros((2, NC), np.complex128)
            for row, w in model64.terms(*cur):
                Y += float(w) * (X[None, :] * table[row]) * D[None, :]
            return Y
        y64 = out_of(spectra(X64, D64))
        X_dev = e.rfft_device(win)[0].astype(np.complex128) / N              # device forward transform (unnormalised)
        D_dev, Y_dev = e.stage_taps(pos[None, :], win)
        D_dev = D_dev[0].astype(np.complex128)
        D_dev[-1] = D_dev[-1].real
        Y_dev = Y_dev[0].astype(np.complex128)
        errs = {"fwd": out_of(spectra(X_dev, D64)) - y64,
                "dist": out_of(spectra(X64, D_dev)) - y64,
                "spec": out_of(Y_dev) - y64,
                "total": y_dev - y64}
        # the filter's own share: the device's spectra against float64 spectra formed from the device's X and D
        errs["filt"] = out_of(Y_dev) - out_of(spectra(X_dev, D_dev))
        if blk == 0:
            continue      # the first block fades in from the constructor's position (0, 0): another computation
        for k, v in errs.items():
            acc[k] += float(np.sum(v ** 2))
        n += y64.size
        worst = max(worst, float(np.abs(errs["total"]).max()))
    e.close()
    r = {k: np.sqrt(v / n) * 1e8 for k, v in acc.items()}
    r["inverse"] = np.sqrt(max(r["total"] ** 2 - r["spec"] ** 2, 0.0))
    rows.append(r)
    print(f"seed {seed}: rms x1e-8  forward {r['fwd']:.2f}  distance factor {r['dist']:.2f}  filter {r['filt']:.2f}  "
          f"-> spectra {r['spec']:.2f}  inverse+crossfade (from total, in quadrature) {r['inverse']:.2f}  total {r['total']:.2f}"
          f"   worst sample {worst:.3e}")
m = {k: np.sqrt(np.mean([r[k] ** 2 for r in rows])) for k in rows[0]}
print("all seeds: " + "  ".join(f"{k} {v:.2f}" for k, v in m.items()))
print("shares of the total error power: " + "  ".join(f"{k} {100 * m[k] ** 2 / m['total'] ** 2:.0f} %" for k in ("fwd", "dist", "filt", "inverse")))
