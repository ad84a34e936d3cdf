#!/usr/bin/env python3
"""Timeline of one bench step from a rocprofv3 kernel_trace.csv: python tools/ktimeline.py <csv> [step_index]
Prints every kernel of the step with its start offset, duration and the idle gap before it."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"np2::(\w+)", n)
    if m: return m.group(1)
    m = re.search(r"_ZNS_\d+(k_[a-z0-9_]+?)E", n) or re.search(r"_ZNS_\d+(k_[a-z0-9_]+)", n)  # (the batched launch template)
    if m: return m.group(1)
    if "init_lookback" in n: return "prim:init"
    m = re.search(r"wrapped_(\w+?)_config", n)
    if m: return "prim:" + m.group(1)
    return n[:28]
marks = [i for i, r in enumerate(rows) if "k_diff_reads" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
a, b = marks[k], marks[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
gaps = []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    busy += (e - s) / 1e3
    gaps.append(gap)
    grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
    wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap:6.1f} gap  {(e - s) / 1e3:7.1f} us  {short(r['Kernel_Name'])}  grid {grid}/{wg}")
    prev_end = max(prev_end, e)
print(f"step span {(prev_end - t0) / 1e3:.1f} us, busy {busy:.1f} us, launches {b - a}, gaps>10us: {sum(g for g in gaps if g > 10):.1f} us, small gaps: {sum(g for g in gaps if 0 < g <= 10):.1f} us")
