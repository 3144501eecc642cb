#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel, counters of its N largest dispatches and totals."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
pick = sys.argv[2] if len(sys.argv) > 2 else "k_trace"
per = defaultdict(dict)
for r in rows:
    if pick not in r["Kernel_Name"]:
        continue
    per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(per)
names = sorted({k for d in per.values() for k in d})
print("dispatches of", pick, len(ids))
print("disp".ljust(8) + "".join(n[:16].rjust(18) for n in names))
for i in ids[:6]:
    print(str(i).ljust(8) + "".join(f"{per[i].get(n, 0):18.4g}" for n in names))
tot = {n: sum(per[i].get(n, 0) for i in ids) for n in names}
print("total".ljust(8) + "".join(f"{tot[n]:18.4g}" for n in names))
