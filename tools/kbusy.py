#!/usr/bin/env python3
"""GPU busy-time analysis of a rocprofv3 kernel trace: union of kernel intervals, concurrency, per-queue load.
usage: python tools/kbusy.py <kernel_trace.csv> [window_ms_from_end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows)
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else None
t_end = max(e[1] for e in ev)
if win:
    ev = [e for e in ev if e[0] >= t_end - win]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]; tot = 0
for s, e, q, n in ev:
    tot += e - s
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
qs = {}
for s, e, q, n in ev:
    qs[q] = qs.get(q, 0) + (e - s)
print(f"window {(t1-t0)/1e6:.2f} ms; kernels {len(ev)}; sum of durations {tot/1e6:.2f} ms; union busy {busy/1e6:.2f} ms "
      f"({busy/(t1-t0)*100:.0f}% of window); mean concurrency while busy {tot/busy:.2f}")
print("per queue busy ms:", {k: round(v/1e6, 2) for k, v in sorted(qs.items())})
