#!/usr/bin/env python3
"""Per kernel, the counters of tools/pmc_units.sh summed over its dispatches and, for the cycle counters, as a share of the
SIMD cycles of the kernel (1024 SIMDs x the kernel's busy cycles; SQ_*ACTIVE*/SQ_*CYCLES* count quad-cycles summed over waves)."""
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]
SIMDS = 1024


def totals(pattern):
    out = {}
    for path in glob.glob(pattern, recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
            if name.startswith("void "):
                name = name[5:]
            name = re.split(r"[<(]", name)[0].strip()
            out.setdefault(name, {}).setdefault(row["Counter_Name"], 0.0)
            out[name][row["Counter_Name"]] += float(row["Counter_Value"])
    return out


res = {}
for which in ("active", "detail", "l1a", "l1b", "l1c"):
    for k, c in totals(os.path.join(root, which, "**", "*_counter_collection.csv")).items():
        if not k.startswith("k_"):
            continue
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8
        e = res.setdefault(k, {})
        for name, v in sorted(c.items()):
            if name == "GRBM_GUI_ACTIVE":
                continue
            e[name] = v
            if gui and ("CYCLES" in name or "ACTIVE" in name or "WAIT" in name or "FULL" in name or "CONFLICT" in name or "STALL" in name or "BUSY" in name or "GATE" in name or "LATENCY" in name):
                if name.startswith("SQ_"):
                    e[name + "_x4_per_simd_cycle"] = round(v * 4 / (SIMDS * gui), 4)
                elif "LATENCY" not in name:
                    e[name + "_per_cu_cycle"] = round(v / (256 * gui), 4)  # (_sum counters: summed over the 256 compute units)
json.dump(res, sys.stdout, indent=1)
