#!/usr/bin/env python3
"""Per-step view of a rocprofv3 --stats kernel_stats.csv: python tools/kstats.py <csv> <steps_incl_warmup>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
out = []
for r in rows:
    n = r["Name"]
    m = re.search(r"np2::(\w+)", n)
    mb = re.search(r"k_np2_batched(?:_wILi\d+ELi|ILi)(\d+)ETnDaXadL_ZN(?:S_|3np2|12_GLOBAL__N_1)*(\d+)(k_[a-zA-Z_0-9]+)", n)
    if mb:  # generic batched kernel template (np2_launch.hpp): the body's name is mangled inside
        ln = int(mb.group(2))
        nm = mb.group(3)[:ln]
        mt = re.search(re.escape(nm) + r"IL[jim](\d+)E", n)
        if mt:
            nm += f"<{mt.group(1)}>"
    elif m:
        nm = m.group(1)
    elif "init_lookback" in n:
        nm = "prim:init_lookback"
    else:
        m = re.search(r"wrapped_(\w+?)_config", n)
        nm = "prim:" + (m.group(1) if m else n[:40])
    us = float(r["TotalDurationNs"]) / steps / 1e3
    tot += us
    out.append((us, nm, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3))
for us, nm, calls, avg in sorted(out, reverse=True):
    print(f"{nm:34s} calls/step {calls:6.1f}  us/step {us:8.1f}  avg {avg:7.1f}")
print(f"total us/step {tot:.1f}; launches/step {sum(o[2] for o in out):.1f}")
