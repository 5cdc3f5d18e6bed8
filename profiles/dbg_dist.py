import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, model64, oracle_lib
from jf_load import jf
hrir = np.load('/root/repo/tests/golden/kemar_hrir_710x2x128_i16.npy').astype(np.float32)/np.float32(32768)
rng = np.random.default_rng(11)
sig = rng.uniform(-.5, .5, 8192).astype(np.float32)
for r in (0.05, 0.5, 1.0, 2.0, 3.5, 4.9):
    e = jf.Engine(256, 512, 1, hrir=hrir); m = model64.Model(256, 512, 1, hrir); o = oracle_lib.Engine(256,512,1,hrir)
    for x in (e, m, o):
        x.set_signal(0, sig); x.set_spherical(0, 0, 45, r)
    for k in range(4):
        y, y64, y32 = e.process_block(), m.process_block(), o.process_block()
        print(r, k, 'hip-64 %.3e  c32-64 %.3e  peak %.3f' % (np.abs(y-y64).max(), np.abs(y32-y64).max(), np.abs(y64).max()))
    e.close()
