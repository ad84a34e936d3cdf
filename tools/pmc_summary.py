#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (counter_collection.csv), e.g.
   python tools/pmc_summary.py gpurun_out/pmc_f/*/*_counter_collection.csv gpurun_out/pmc_w/*/*_counter_collection.csv
Prints JSON: {kernel: {counter_avg_per_launch: value}}.  FETCH_SIZE / WRITE_SIZE are reported in KB."""
import csv, json, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        mb = re.search(r"k_np2_batched(?:_wILi\d+ELi|ILi)(\d+)ETnDaXadL_ZN(?:S_|3np2|12_GLOBAL__N_1)*(\d+)(k_[a-zA-Z_0-9]+)", n)
        m = re.search(r"np2::(\w+)", n)
        name = "np2::" + (mb.group(3)[:int(mb.group(2))] if mb else m.group(1)) if (mb or m) else n[:48]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c + "_KB_avg_per_launch": round(sum(v) / len(v), 1) for c, v in cs.items()} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
