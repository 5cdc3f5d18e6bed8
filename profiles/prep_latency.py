"""prep_kernel at four grid sizes under rocprofv3 --kernel-trace: the chain of one wave against the full launch
(cd /tmp; rocprofv3 --kernel-trace --output-format csv -d out -- python3 profiles/prep_latency.py)."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from jf_load import jf
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
hrir = np.load(os.path.join(ROOT, "tests", "golden", "kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
for S, K in ((4, 2), (64, 8), (1024, 16), (1024, 128)):
    e = jf.Engine(256, 512, S, hrir=hrir, max_batch_blocks=K)
    e.set_prep_ahead(False)
    rng = np.random.default_rng(1)
    for s in range(S):
        e.set_signal(s, rng.uniform(-.5, .5, 4096).astype(np.float32))
    ele = rng.uniform(-40, 80, (2 * K, S)).astype(np.float32); azi = rng.uniform(0, 359, (2 * K, S)).astype(np.float32)
    pos = jf.positions_from_spherical(ele, azi, np.full((2 * K, S), 1.0, np.float32))
    e.upload_positions(pos)
    for i in range(20):
        e.batch_run((i % 2) * K, K)
    e.synchronize()
    e.close()
