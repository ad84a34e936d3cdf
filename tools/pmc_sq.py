#!/usr/bin/env python3
"""Per-kernel averages of SQ counters from rocprofv3 --pmc passes: python tools/pmc_sq.py <counter_collection.csv>..."""
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        mb = re.search(r"k_np2_batched(?:_wILi\d+ELi|ILi)(\d+)ETnDaXadL_ZN(?:S_|3np2|12_GLOBAL__N_1)*(\d+)(k_[a-zA-Z_0-9]+)", n)
        m = re.search(r"np2::(\w+)", n)
        name = mb.group(3)[:int(mb.group(2))] if mb else (m.group(1) if m else n[:40])
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = sorted({c for cs in acc.values() for c in cs})
print("kernel".ljust(28), " ".join(c[-14:].rjust(14) for c in cols))
for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0])) / max(1, len(kv[1].get("SQ_BUSY_CYCLES", [0])))):
    print(k[:28].ljust(28), " ".join(("%14.0f" % (sum(cs[c]) / len(cs[c])) if c in cs else " " * 14) for c in cols))
