#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and the per-launch durations of one batch."""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\((v4r::)?(SceneDev|\(anonymous).*", "", n)
    return n.replace("void ", "")[:44]


rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
tot = defaultdict(lambda: [0, 0.0])
for s, e, n in ks:
    tot[n][0] += 1
    tot[n][1] += (e - s) / 1e3
span = (ks[-1][1] - ks[0][0]) / 1e3
print(f"span {span/1e3:.2f} ms, kernels busy {sum(v[1] for v in tot.values())/1e3:.2f} ms")
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:44s} calls {c:6d} total {us/1e3:9.2f} ms avg {us/c:9.2f} us")
# a batch starts with the trace launch that generates the camera rays (last template argument true)
idx = [i for i, k in enumerate(ks) if re.match(r"k_trace<\w+, \d+, \w+, true, false(, \d+)?>$", k[2])]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
if len(idx) > which + 1:
    i0, i1 = idx[which], idx[which + 1]
    tr = [round((e - s) / 1e3, 1) for s, e, n in ks[i0:i1] if "k_trace" in n or "k_tail" in n]
    print(" trace launches in order (first launch, then [second launch] of a segment):",
          " ".join(("[%s]" if re.match(r"k_trace<\w+, \d+, \w+, \w+, true", n) else "tail:%s" if "k_tail" in n else "%s") % round((e - s) / 1e3, 1)
                   for s, e, n in ks[i0:i1] if "k_trace" in n or "k_tail" in n))
    print("batch", which, "total", round((ks[i1][0] - ks[i0][0]) / 1e3, 1), "us; launches", i1 - i0)
    print(" trace us:", tr[:16], "... tail avg", round(sum(tr[16:]) / max(1, len(tr[16:])), 1), "sum", round(sum(tr), 1))
    for cls in sorted({n for _, _, n in ks[i0:i1] if "k_shade" in n}):
        sh = [round((e - s) / 1e3, 1) for s, e, n in ks[i0:i1] if n == cls]
        print(f" {cls}:", sh[:10], "... sum", round(sum(sh), 1))
    gaps = sum(ks[i + 1][0] - ks[i][1] for i in range(i0, i1 - 1)) / 1e3
    if gaps < 0:
        print(" (kernels of several streams overlap in this trace: the per-batch view above mixes batches; profile with V4R_PIPES=1 for a clean timeline)")
    else:
        print(" idle gaps between kernels:", round(gaps, 1), "us")
