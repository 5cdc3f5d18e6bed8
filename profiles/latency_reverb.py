"""Real-time (one block per call) cost of configs[4]: 256 sources, B = 128, 2 s IR, reverb + spatialiser, per block of the
16-block cycle of the non-uniformly partitioned stage (and, with JF_RV_UNIFORM=1, of uniform partitioning: round 3's form).
JF_RV_CALLS=n: n timed calls (default 1600; a multiple of 16).  JF_RV_PACED_US=2902: the calls at the audio callback's own
cadence -- one per block period, the thread spinning in between -- instead of back to back (round 6): what a PortAudio host sees;
the side stream's work of a big-block boundary then has the whole period to itself."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jf_load import jf
if not os.environ.get("JF_NO_PIN"):
    jf.pin_thread_to_device(0)   # the audio thread on the GPU's NUMA node (include/jefferson.h; profiles/r04/rt_numa.md)
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
rng = np.random.default_rng(99)
ir = rng.standard_normal(88200) * np.exp(-6.9 * np.arange(88200) / 88200.0)
ir = (ir / np.sqrt((ir ** 2).sum())).astype(np.float32)
S, B = int(os.environ.get("JF_RV_SOURCES", "256")), 128
for uniform in ((False,) if os.environ.get("JF_RV_ONLY_NONUNIFORM") else (False, True)):
    e = jf.Engine(B, 512, S, hrir=hrir)
    e.set_reverb_partitioning(1 if uniform else 0)
    for s in range(S):
        e.set_signal(s, np.random.default_rng(1234 + s).uniform(-.5, .5, 44100).astype(np.float32))
        e.set_spherical(s, -40 + (s * 7) % 121, (s * 37) % 360, 1.0)
    if os.environ.get("JF_RV_NO_AHEAD"):      # round 4's order: every call runs its own stage in front of its spatialiser
        e.set_reverb_ahead(0)
    if os.environ.get("JF_RV_HEAD_OWN"):      # round 4's form: the head as a kernel of its own in front of the real-time kernel
        e.set_reverb_head_fused(0)
    if os.environ.get("JF_RV_SIDE_WGS"):      # workgroups of the side stream's product kernel (jefferson_debug.h)
        e.set_reverb_side_workgroups(int(os.environ["JF_RV_SIDE_WGS"]))
    e.set_reverb(ir, 0.5)
    out = np.zeros(2 * B, np.float32)
    fp = out.ctypes.data_as(jf._f)
    L = jf.lib()
    for k in range(64):
        L.jf_process_block(e.h, fp)
    ts = []
    n_calls = int(os.environ.get("JF_RV_CALLS", "1600")) // 16 * 16
    paced = float(os.environ.get("JF_RV_PACED_US", "0")) * 1e-6
    t_next = time.perf_counter()
    for k in range(64, 64 + n_calls):
        if paced:
            t_next += paced
            while time.perf_counter() < t_next:
                pass
        if k % 7 == 0:      # sources move now and then, as they would
            for s in range(0, S, 5):
                e.set_spherical(s, -40 + (s * 7) % 121, (s * 37 + k) % 360, 1.0)
        t0 = time.perf_counter(); L.jf_process_block(e.h, fp); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    n, head, big, taps = e.reverb_partitions()
    print(f"configs[4] real-time, {S} sources, B = 128, {n} partitions of 128 as {head} x 128" + (f" + {big} x {taps}" if big else "")
          + f": jf_process_block mean {ts.mean():.1f} us, median {np.median(ts):.1f}, p99 {np.percentile(ts, 99):.1f}, max {ts.max():.1f}"
          f" ({len(ts)} calls " + (f"at one per {paced * 1e6:.0f} us" if paced else "back to back") + "; block period 2902 us)")
    if big:
        cyc = ts.reshape(-1, 16)      # block j of the cycle: j = 15 forms X_m behind its head, j = 0 forms TAIL(m) in front of it
        print("   by place in the 16-block cycle (median us):", " ".join(f"{v:.0f}" for v in np.median(cyc, axis=0)))
    e.close()
