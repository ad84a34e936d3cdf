#!/usr/bin/env python3
"""Where a traced run spent its wall clock: reads rocprofv3's kernel_trace / hip_api_trace CSVs under a directory and prints the longest
kernels, the longest HIP API calls and the longest stretches with no kernel running, all as offsets from the first event."""
import csv, glob, os, sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12


def rows(pat):
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        with open(f, newline="") as fh:
            yield from csv.DictReader(fh)


ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in rows("*kernel_trace.csv")]
hs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "?")) for r in rows("*hip_api_trace.csv")]
if not ks and not hs:
    sys.exit("no trace rows under " + d)
t0 = min([k[0] for k in ks] + [h[0] for h in hs])
ms = lambda t: (t - t0) / 1e6
print(f"{len(ks)} kernels, {len(hs)} HIP calls; span {ms(max([k[1] for k in ks] + [h[1] for h in hs])):.0f} ms")
print("longest kernels (start ms, duration ms):")
for s, e, n in sorted(ks, key=lambda k: k[0] - k[1])[:top]:
    print(f"  {ms(s):9.1f} {(e - s) / 1e6:9.2f}  {n}")
print("longest HIP calls (start ms, duration ms, thread):")
for s, e, n, th in sorted(hs, key=lambda h: h[0] - h[1])[:top * 2]:
    print(f"  {ms(s):9.1f} {(e - s) / 1e6:9.2f}  {n}  [{th}]")
ks.sort()
gaps, end = [], None
for s, e, n in ks:
    if end is not None and s > end:
        gaps.append((s - end, end, n))
    end = e if end is None else max(end, e)
print("longest stretches with no kernel running (start ms, length ms, next kernel):")
for g, at, n in sorted(gaps, reverse=True)[:top]:
    print(f"  {ms(at):9.1f} {g / 1e6:9.2f}  {n}")
