"""Driver for the LDS bank-conflict attribution (profiles/lds_attribution.sh): runs the two debug kernels that execute the
fused kernel's device code in isolation -- rfft_debug_kernel (the forward transform: pass A/B/C exchanges + the mirror)
and stage_debug_kernel (distance factors from the twiddle table, forward transform, weighted spectra) -- on many windows,
so that rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS attributes the conflicts per stage."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jf_load import jf  # noqa: E402

hrir = np.load(os.path.join(ROOT, "tests", "golden", "kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
rng = np.random.default_rng(0)
N = 8192
e = jf.Engine(256, 512, 1, hrir=hrir)
win = rng.uniform(-0.5, 0.5, (N, 1024)).astype(np.float32)
for _ in range(3):
    e.rfft_device(win)
# the bench's radius distribution: r uniform in [0.5, 3.5] through the spherical setter
pos = np.zeros((N, 5), np.float32)
for i in range(N):
    pos[i] = jf.position_from_spherical(-40 + (7 * i) % 121, (37 * i) % 360, 0.5 + 3.0 * rng.random())
for _ in range(3):
    e.stage_taps(pos, None)        # distance factors only
for _ in range(3):
    e.stage_taps(pos, win)         # + forward transform + weighted spectra (filtered_bins)
e.close()
print("done")
