#!/usr/bin/env python3
"""The copy / fill / marker launches of a rocprofv3 kernel trace with their sizes: python tools/kcopies.py <kernel_trace.csv> <steps>"""
import csv, sys, collections
sys.path.insert(0, "tools")
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
import re
def short(n):
    mb = re.search(r"k_np2_batched(?:_wILi\d+ELi|ILi)(\d+)ETnDaXadL_ZN(?:S_|3np2|12_GLOBAL__N_1)*(\d+)(k_[a-zA-Z_0-9]+)", n)
    if mb: return mb.group(3)[:int(mb.group(2))]
    m = re.search(r"np2::(\w+)", n)
    return m.group(1) if m else n[:40]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = short(r["Kernel_Name"])
    if n not in ("k_copy", "k_fill", "k_post", "k_flush_done", "k_copy_counted"): continue
    g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
    a = agg[(n, g)]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (n, g), (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:40]:
    print(f"{n:16s} grid_threads {g:10d} (~{g*16/1e6:8.2f} MB at 16 B/thread)  {c/steps:6.2f}/step  {t/steps:8.1f} us/step  avg {t/c:7.1f}")
