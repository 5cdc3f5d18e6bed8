#!/usr/bin/env python3
"""Static ISA statistics of one kernel from hipcc -S output: instruction counts by class, code bytes (8 B for VOP3/VOP3P/
literal forms is not modelled: counts only), v_readlane/v_writelane (SGPR spill traffic).
usage: isa_stats.py file.s mangled-kernel-substring"""
import re, sys
txt = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
on = False
cnt = {}
for l in txt:
    if re.match(r"^_Z\w*%s\w*:\s*(;.*)?$" % re.escape(key), l):
        on = True
        continue
    if on and l.startswith(".Lfunc_end"):
        break
    if not on:
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", l + " ")
    if not m:
        continue
    op = m.group(1)
    cls = ("valu_pk" if op.startswith("v_pk_") else "lane" if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32") else
           "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith("s_load") and not op.startswith("s_waitcnt") else
           "smem" if op.startswith("s_load") else "waitcnt" if op.startswith("s_waitcnt") else "lds" if op.startswith("ds_") else
           "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    cnt[cls] = cnt.get(cls, 0) + 1
    if op in ("v_readlane_b32", "v_writelane_b32"):
        cnt[op] = cnt.get(op, 0) + 1
print(" ".join(f"{k}={v}" for k, v in sorted(cnt.items())), "total=%d" % sum(v for k, v in cnt.items() if not k.startswith("v_r") and not k.startswith("v_w")))
