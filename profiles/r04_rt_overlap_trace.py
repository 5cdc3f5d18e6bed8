"""From a rocprofv3 --kernel-trace csv of profiles/r04_rt_async_ab.py: per block, when the spatialiser starts and ends relative
to the start of the reverb's head kernel (ns)."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
head = [r for r in rows if "reverb_mac_kernel" in r["Kernel_Name"]]
rt = [r for r in rows if "rt_block_kernel" in r["Kernel_Name"]]
n = min(len(head), len(rt))
h0 = np.array([int(r["Start_Timestamp"]) for r in head[-n:]]); h1 = np.array([int(r["End_Timestamp"]) for r in head[-n:]])
r0 = np.array([int(r["Start_Timestamp"]) for r in rt[-n:]]); r1 = np.array([int(r["End_Timestamp"]) for r in rt[-n:]])
sel = slice(n // 2, n)
print(f"{n} blocks: head duration median {np.median((h1 - h0)[sel]):.0f} ns; spatialiser starts {np.median((r0 - h0)[sel]):.0f} ns after the head's "
      f"start, {np.median((r0 - h1)[sel]):.0f} after its end; spatialiser duration {np.median((r1 - r0)[sel]):.0f}; head start to spatialiser end "
      f"{np.median((r1 - h0)[sel]):.0f}; block to block {np.median(np.diff(h0)[sel]):.0f}")
