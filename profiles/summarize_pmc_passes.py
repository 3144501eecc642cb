#!/usr/bin/env python3
"""Fold the rocprofv3 passes over one single-stream bench.py step into one JSON: per kernel, HBM bytes from the
counters over the kernel's exclusive time (fraction of the 8 TB/s HBM peak), and for the traversal kernels the
vector-issue side (fraction of the SIMDs' issue slots used, lanes active per instruction, where the waves wait).

    python profiles/summarize_pmc_passes.py <dir> <workload> > out.json

<dir> holds pmc2_{fetch,write,l2,sq,ta}/*_counter_collection.csv (one counter set per pass; gfx950 fits no more) and
pmc2_time/kernel_trace.csv (the same command under --kernel-trace, no counters: the kernels' durations):
    FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum |
    SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
    SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE | TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_WAIT_ANY
    SQ_BUSY_CYCLES
Every pass runs with V4R_PIPES=1, so the kernels of a step run one after the other and a duration is exclusive.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

HBM_PEAK = 8.0e12       # B/s
SIMDS = 256 * 4         # 256 CUs x 4 SIMDs
VALU_ISSUE_CYCLES = 2   # a wave64 vector instruction occupies its SIMD's issue port for 2 cycles (32 lanes per cycle)
SIZES = {"c1": (800, 800), "c2": (1920, 1080), "c3": (1024, 1024), "c4": (1920, 1080), "c5": (3840, 2160)}


SHADE_CLASS = {"0": "light", "1": "lambert", "2": "lambert_noise", "3": "metal", "4": "dielectric", "5": "isotropic"}


def short(name):
    k = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    m = re.match(r"k_shade<(\d)", k)
    if m:  # one entry per material class: their costs differ by an order of magnitude (Perlin noise)
        return "k_shade_" + SHADE_CLASS.get(m.group(1), m.group(1))
    return re.sub(r"[<(].*", "", k)


def kernel_sources_sha256():
    """Hash of the device sources the profiled library was built from; bench.py compares it with the sources it runs."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    for rel in ("ray-tracing_amd/csrc/device/v4r_device.hip", "ray-tracing_amd/csrc/device/v4r_scene_dev.hpp",
                "ray-tracing_amd/csrc/device/v4r_math.hpp", "include/v4ray_hip.h"):
        with open(os.path.join(here, "..", rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def totals(pattern):
    t = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    for path in glob.glob(pattern):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            t[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    return t, {k: len(v) for k, v in n.items()}


def durations(path):
    t = defaultdict(float)
    for r in csv.DictReader(open(path)):
        t[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return t


root, workload = sys.argv[1], sys.argv[2]
spp = 64
w_, h_ = SIZES[workload]
paths = w_ * h_ * spp
f, nf = totals(os.path.join(root, "pmc2_fetch", "*_counter_collection.csv"))
w, _ = totals(os.path.join(root, "pmc2_write", "*_counter_collection.csv"))
l2, _ = totals(os.path.join(root, "pmc2_l2", "*_counter_collection.csv"))
sq, _ = totals(os.path.join(root, "pmc2_sq", "*_counter_collection.csv"))
ta, _ = totals(os.path.join(root, "pmc2_ta", "*_counter_collection.csv"))
secs = durations(os.path.join(root, "pmc2_time", "kernel_trace.csv"))
out = {"command": f"V4R_PIPES=1 rocprofv3 --pmc <one counter set per pass> --output-format csv -- python3 bench.py --workload {workload} "
                  "--steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-side --spp 64   (+ one --kernel-trace pass for the durations)",
       "workload": f"{workload} {w_}x{h_}, {spp} spp", "paths": paths, "kernel_sources_sha256": kernel_sources_sha256(),
       "note": "FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so read bytes = 2 x "
               "FETCH_SIZE x 1024. SQ_*_CYCLES count quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs. "
               "hbm_frac = HBM bytes / exclusive kernel seconds / 8 TB/s. valu_issue_frac = SQ_INSTS_VALU x 2 cycles / "
               "(1024 SIMDs x kernel cycles): the share of the vector issue slots the kernel used. resident_waves_per_simd = "
               "SQ_WAVE_CYCLES x 4 / (1024 SIMDs x kernel cycles): the occupancy the launches reached on average (of 8 possible).",
       "kernels": {}}
tot_bytes = tot_secs = 0.0
for k in sorted(f):
    if not k.startswith("k_"):
        continue
    rd, wr = 2 * f[k]["FETCH_SIZE"] * 1024, w[k]["WRITE_SIZE"] * 1024
    hit, miss = l2[k]["TCC_HIT_sum"], l2[k]["TCC_MISS_sum"]
    t = secs.get(k, 0.0)
    e = {"dispatches": nf[k], "seconds": round(t, 6), "hbm_read_bytes": rd, "hbm_write_bytes": wr,
         "hbm_bytes_per_path": round((rd + wr) / paths, 1),
         "hbm_GBps": round((rd + wr) / t / 1e9, 1) if t else None,
         "hbm_frac": round((rd + wr) / t / HBM_PEAK, 4) if t else None,
         "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None}
    tot_bytes += rd + wr
    tot_secs += t
    s, a = sq[k], ta[k]
    if s.get("SQ_INSTS_VALU"):
        wc, gui = s["SQ_WAVE_CYCLES"], s["GRBM_GUI_ACTIVE"] / 8
        e.update({"valu_insts": s["SQ_INSTS_VALU"], "salu_insts": s["SQ_INSTS_SALU"], "vmem_rd_insts": s["SQ_INSTS_VMEM_RD"],
                  "lds_insts": s["SQ_INSTS_LDS"],
                  "valu_insts_per_path": round(s["SQ_INSTS_VALU"] * 64 / paths, 1),
                  "active_lanes_per_valu_inst": round(s["SQ_THREAD_CYCLES_VALU"] / s["SQ_INSTS_VALU"], 1),
                  "valu_issue_frac": round(s["SQ_INSTS_VALU"] * VALU_ISSUE_CYCLES / (SIMDS * gui), 4),
                  "wave_time_issuing": round(s["SQ_ACTIVE_INST_ANY"] / wc, 3),
                  "wave_time_issue_stalled": round(s["SQ_WAIT_INST_ANY"] / wc, 3),
                  "wave_time_waiting_on_memory_or_barrier": round(a["SQ_WAIT_ANY"] / wc, 3),
                  "resident_waves_per_simd": round(wc * 4 / (SIMDS * gui), 2),
                  "kernel_cycles": gui,
                  "clock_GHz": round(gui / t / 1e9, 3) if t else None,
                  "ta_busy": round(a["TA_TA_BUSY_sum"] / (256 * gui), 3),
                  "l1_hit_rate": round(1 - a["TCP_TCC_READ_REQ_sum"] / a["TCP_TOTAL_CACHE_ACCESSES_sum"], 3)})
    out["kernels"][k] = e
out["all_kernels"] = {"seconds": round(tot_secs, 6), "hbm_bytes_per_path": round(tot_bytes / paths, 1),
                      "hbm_frac": round(tot_bytes / tot_secs / HBM_PEAK, 4) if tot_secs else None,
                      "Mpaths_per_s_single_stream": round(paths / tot_secs / 1e6, 1) if tot_secs else None}
json.dump(out, sys.stdout, indent=1)
