#!/usr/bin/env python3
"""Where a rocprofv3 kernel trace's idle time is: (1) gaps of the UNION of all queues' kernels (the device has nothing to
run), (2) gaps per queue (a batch group waits for its host thread), both aggregated by the kernel that ended before the
gap and the one that started after it.
usage: python tools/kgaps.py <kernel_trace.csv> [window_ms_from_end] [steps_in_window]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
import re
def short(n):
    mb = re.search(r"k_np2_batched(?:_wILi\d+ELi|ILi)(\d+)ETnDaXadL_ZN(?:S_|3np2|12_GLOBAL__N_1)*(\d+)(k_[a-zA-Z_0-9]+)", n)
    if mb: return mb.group(3)[:int(mb.group(2))]
    m = re.search(r"np2::(\w+)", n)
    return m.group(1) if m else n[:40]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), short(r["Kernel_Name"])) for r in rows)
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else None
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
t_end = max(e[1] for e in ev)
if win: ev = [e for e in ev if e[0] >= t_end - win]
# (1) union gaps
gaps = collections.defaultdict(lambda: [0, 0])
cur_e, last = ev[0][1], ev[0][3]
tot_gap = 0
for s, e, q, n in ev[1:]:
    if s > cur_e:
        g = gaps[(last, n)]; g[0] += 1; g[1] += s - cur_e; tot_gap += s - cur_e
    if e > cur_e: cur_e, last = e, n
print(f"union idle: {tot_gap/1e3/steps:.1f} us/step over {sum(g[0] for g in gaps.values())/steps:.1f} gaps/step")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda x: -x[1][1])[:25]:
    print(f"  {a:>40} -> {b:<40} {c/steps:6.1f}/step {t/1e3/steps:8.1f} us/step  avg {t/1e3/c:6.1f}")
# (2) per queue
pq = collections.defaultdict(list)
for s, e, q, n in ev: pq[q].append((s, e, n))
qg = collections.defaultdict(lambda: [0, 0]); tq = 0
for q, l in pq.items():
    for (s0, e0, n0), (s1, e1, n1) in zip(l, l[1:]):
        if s1 - e0 > 15000:  # (> 15 us: a host round trip, not a back-to-back dispatch)
            g = qg[(n0, n1)]; g[0] += 1; g[1] += s1 - e0; tq += s1 - e0
print(f"per-queue gaps > 15 us: {tq/1e3/steps:.1f} us/step summed over {len(pq)} queues")
for (a, b), (c, t) in sorted(qg.items(), key=lambda x: -x[1][1])[:30]:
    print(f"  {a:>40} -> {b:<40} {c/steps:6.1f}/step {t/1e3/steps:8.1f} us/step  avg {t/1e3/c:6.1f}")
