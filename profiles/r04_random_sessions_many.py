"""The random sessions of tests/test_gpu_random_sessions.py over many more seeds than the suite carries (one-off soak)."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
from jf_load import jf
import test_gpu_random_sessions as T
hrir = np.load(os.path.join(ROOT, "tests/golden/kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / np.float32(32768)
castanets = (np.load(os.path.join(ROOT, "tests/golden/castanets_441_excerpt_i24.npy")) / 8388608.0).astype(np.float32)
n0, n1 = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(n0)
fails = 0
t0 = time.time()
for seed in range(n0, n1):
    B = int(rng.choice([64, 128, 192, 256]))
    S = int(rng.choice([64, 100, 257, 520] if os.environ.get("JF_SESS_BIG") else [1, 2, 3, 5, 8, 9, 16, 17, 33]))
    reverb = 0 if B == 192 or rng.random() < 0.5 else int(rng.choice([700, 16 * B * 3 + 3, (16 if B <= 128 else 8) * B * 5 + 100]))
    for name, fn, args in (("block+batch", T.test_random_session_of_block_and_batch_calls, (seed, B, S, reverb)),
                           ("callback", T.test_random_session_through_the_callback, (seed, B, S, reverb)),
                           ("trajectory", T.test_random_session_of_runs_over_an_uploaded_trajectory, (seed, B, 4 * max(1, S // 4), int(rng.choice([0, 1, 2, 4])))),
                           ("group", T.test_random_session_through_the_group_of_shards, (seed, B, max(S, 4), int(rng.choice([2, 3, 4])), reverb))):
        try:
            fn(jf, hrir, castanets, *args)
        except AssertionError as ex:
            msg = str(ex).split("\n")[0][:200]
            if ("blocks >" in msg or "peak >" in msg or "prepared >" in msg or "(blocks, peak)" in msg) and "step" not in msg:
                # the sanity counts at a test's end (enough blocks, loud enough, enough prepared runs) depend on the draw:
                # reported, not counted as a failure of the engine
                print("SANITY", name, args, msg, flush=True)
                continue
            print("FAIL", name, args, msg, flush=True)
            traceback.print_exc(limit=1)
            fails += 1
        except Exception as ex:
            print("ERROR", name, args, repr(ex)[:300], flush=True)
            fails += 1
    if (seed - n0) % 10 == 9:
        print(f"seed {seed}: {fails} failures so far, {time.time() - t0:.0f} s", flush=True)
print("DONE", n1 - n0, "seeds,", fails, "failures")
