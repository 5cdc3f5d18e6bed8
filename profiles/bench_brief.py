"""Prints the few numbers of a bench.py JSON line that matter while tuning: python profiles/bench_brief.py <file>..."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as ex:  # an empty or truncated file: say so instead of a traceback
        print(f"{f}: FAILED ({type(ex).__name__}: {ex})")
        continue
    r = d["roofline"]
    iss = r.get("issue") or {}
    print(f"{f}: value {d['value']:.4g}  step {d['ms_per_step']:.4f} ms  kernel {r['kernel']} {r['avg_launch_ms']:.4f} ms  "
          f"frac {r['frac']:.3f}  verified {d.get('verified')}  valu/item {iss.get('valu_insts_per_source_block', 0):.0f}  "
          f"valu-pipe {iss.get('valu_pipe_busy_share_at_2_cycles_per_inst', 0):.2f}  vmem/item {iss.get('vmem_loads_per_source_block', 0):.1f}  "
          f"wait {iss.get('wave_time_waiting_share', 0):.2f}  hbm {(r.get('hbm') or {}).get('frac')}")
