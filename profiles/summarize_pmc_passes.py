#!/usr/bin/env python3
"""Fold the five rocprofv3 --pmc passes over one bench.py command into one JSON (per kernel: HBM bytes, L2/L1 hit
rates, instruction counts, where the waves' time goes).

    python profiles/summarize_pmc_passes.py <dir with pmc2_{fetch,write,l2,sq,ta}/*_counter_collection.csv> <paths> > out.json

The passes (one counter set each; gfx950 fits no more per pass):
    FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum |
    SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
    SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE | TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_WAIT_ANY
    SQ_BUSY_CYCLES
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def totals(pattern):
    t = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    for path in glob.glob(pattern):
        for r in csv.DictReader(open(path)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"[<(].*", "", k).replace("void ", "")
            t[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    return t, {k: len(v) for k, v in n.items()}


root, paths = sys.argv[1], int(sys.argv[2])
f, nf = totals(os.path.join(root, "pmc2_fetch", "*_counter_collection.csv"))
w, _ = totals(os.path.join(root, "pmc2_write", "*_counter_collection.csv"))
l2, _ = totals(os.path.join(root, "pmc2_l2", "*_counter_collection.csv"))
sq, _ = totals(os.path.join(root, "pmc2_sq", "*_counter_collection.csv"))
ta, _ = totals(os.path.join(root, "pmc2_ta", "*_counter_collection.csv"))
out = {"command": "rocprofv3 --pmc <one counter set per pass> --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
                  "--no-cpu-baseline --no-profile --spp 64",
       "workload": "c2 1920x1080, 64 spp", "paths": paths,
       "note": "FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so read bytes = 2 x "
               "FETCH_SIZE x 1024. SQ_*_CYCLES count quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs.",
       "kernels": {}}
for k in sorted(f):
    if not k.startswith("k_"):
        continue
    rd, wr = 2 * f[k]["FETCH_SIZE"] * 1024, w[k]["WRITE_SIZE"] * 1024
    hit, miss = l2[k]["TCC_HIT_sum"], l2[k]["TCC_MISS_sum"]
    e = {"dispatches": nf[k], "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_path": round((rd + wr) / paths, 1),
         "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None}
    s, t = sq[k], ta[k]
    if s.get("SQ_INSTS_VALU"):
        wc, gui = s["SQ_WAVE_CYCLES"], s["GRBM_GUI_ACTIVE"] / 8
        e.update({"valu_insts": s["SQ_INSTS_VALU"], "salu_insts": s["SQ_INSTS_SALU"], "vmem_rd_insts": s["SQ_INSTS_VMEM_RD"],
                  "lds_insts": s["SQ_INSTS_LDS"],
                  "active_lanes_per_valu_inst": round(s["SQ_THREAD_CYCLES_VALU"] / s["SQ_INSTS_VALU"], 1),
                  "wave_time_issuing": round(s["SQ_ACTIVE_INST_ANY"] / wc, 3),
                  "wave_time_issue_stalled": round(s["SQ_WAIT_INST_ANY"] / wc, 3),
                  "wave_time_waiting_on_memory_or_barrier": round(t["SQ_WAIT_ANY"] / wc, 3),
                  "kernel_cycles": gui, "ta_busy": round(t["TA_TA_BUSY_sum"] / (256 * gui), 3),
                  "l1_hit_rate": round(1 - t["TCP_TCC_READ_REQ_sum"] / t["TCP_TOTAL_CACHE_ACCESSES_sum"], 3)})
    out["kernels"][k] = e
json.dump(out, sys.stdout, indent=1)
