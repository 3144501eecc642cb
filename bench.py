#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s of the render-loop hot path on N MI355X GPUs of one node.

Workload (config.workload): by default the bunny scene (c4: bunny.obj subdivided to 79 488 triangles + an
earth-textured sphere in a Cornell box) at 1920x1080, 512 samples per pixel per GPU -- the scene the
>= 100x GPU/CPU target is stated on.  One "step" renders those 512 full-image 1-spp passes on every rank
(rank r takes passes r, r+N, r+2N, ... of one N*512-spp job, so per-GPU work is fixed: weak scaling) and, for
N > 1, sums the per-GPU radiance accumulators onto rank 0 with one RCCL reduce over xGMI, ordered on the
renderer's own stream.  Scene tables are resident in HBM before the timed region starts; nothing crosses
PCIe inside it.  For N > 1 a second leg (`c5_job`) times the job the multi-GPU target names: zy_scene.json at
3840x2160, 4096 passes in total split over the ranks by sample range (strong scaling), render and reduce apart.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.

`roofline` prices the kernel group with the largest exclusive device time (BVH traversal + intersection on every
configured scene but the 4K textured-spheres one, where shading weighs as much).  After the timed
region the same renderer renders 64 more passes with every batch on one stream and HIP events around the
traversal, shading and resolve launches (exclusive kernel times), then the same passes once more with the
instrumented kernels (exact work counts).  From those:
  * hbm: the bytes a kernel group must move through HBM (its queue records: rays, hit records, class-queue
    indices, radiance slots -- counted from the work counters x the record sizes) per launch, over its average
    launch duration, against the 8 TB/s peak; `traffic` is the HBM bytes per launch that rocprofv3's
    FETCH_SIZE / WRITE_SIZE counters measured for the same workload (committed under profiles/), scaled to
    this run's launch count.  The BVH nodes and primitive records a ray touches are table reads served by
    LDS / L1 / L2 (the tables are <= 15 MB) and are listed apart (`tables`), not priced against HBM.
  * issue: vector instructions issued (SQ_INSTS_VALU of the committed counter passes, per path) x 2 cycles
    over the SIMD cycles of the measured kernel time at the 2.4 GHz peak clock, and the lanes active per
    vector instruction.
Both fractions of the dominant group are in the block (`hbm_traffic_frac`, `issue_frac`); `bound` names the larger and
`frac` repeats it.  The hardware counters are not collected in this run: they come from the committed summary of the
same workload (`pmc.source`), tied to the kernels by a hash of the device sources (`pmc.kernel_sources_match`).
`cpu_baseline` is the CPU oracle (the scalar C++ restatement of the reference's loop, oracle/) timed on every
hardware thread this host offers, on a bounded sample of the same workload -- a reported baseline, not the
optimisation target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing_amd"))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak
PEAK_CLOCK_HZ = 2.4e9
SIMDS = 256 * 4
VALU_ISSUE_CYCLES = 2       # a wave64 vector instruction occupies its SIMD's issue port for 2 cycles
# Queue record sizes (bytes): ray record, hit record + class-queue index, radiance slot; table records: wide BVH
# node, primitive, triangle shading record, texel
RAY_RECORD, HIT_RECORD, QUEUE_INDEX, RADIANCE, NODE, PRIM, TRI_SHADING, TEXEL = 48, 16, 4, 16, 128, 48, 64, 4
PROFILE_PASSES = 64
PMC_SUMMARY = {w: f"profiles/r03_{w}_pmc.json" for w in ("c1", "c2", "c3", "c4", "c5")}

WORKLOADS = {
    "c1": "data/scene.json verbatim (Cornell walls, 4 968-triangle bunny, fog sphere, 4 985 shapes)",
    "c2": "Book-1 random-spheres scene (484 spheres, Lambertian/metal/dielectric)",
    "c3": "Cornell box, cosine + light-list mixture sampling",
    "c4": "bunny.obj subdivided to 79 488 triangles + earth-textured sphere in a Cornell box",
    "c5": "zy_scene.json (noise/image-textured spheres in a Cornell box)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="passes per GPU per step (default: the workload's full count)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the exclusive-timing and work-counter legs (no roofline)")
    ap.add_argument("--split", default="sample", choices=["sample", "tile"],
                    help="N > 1: rank r renders passes r, r+N, ... of the job (sample) or all passes of its band of rows (tile)")
    ap.add_argument("--c4-subdivide", type=int, default=2, help="midpoint subdivisions of bunny.obj in c4 (2: 79 488 triangles, 0: 4 968)")
    ap.add_argument("--flags", type=int, default=0, help="scene flags (experiments: 4 = per-ray plane light pdf, 1 = bounded light pdf)")
    ap.add_argument("--no-side", action="store_true", help="skip the short random-spheres (c2) measurement added to the c4 line")
    ap.add_argument("--c5-spp", type=int, default=4096,
                    help="N > 1: total passes of the second leg, the 3840x2160 textured-spheres job split over the ranks (0 = skip the leg)")
    return ap.parse_args()


def queue_bytes(stats):
    """Bytes each kernel group must move through HBM for the counted work (queue records only).
    k_trace: per ray segment the 32 B of a ray record that hold origin, direction and path slot (read from the queue; a
    camera ray is generated in registers and its 48-B record written instead, when it hits), per queued hit a 16-B hit
    record + a 4-B class-queue index written, per miss or hit on a plain light a 16-B radiance slot written and the
    ray's throughput read.  k_shade: per queued hit the index, the ray and the hit record
    read, then the scattered ray (48 B) or, when the path ends there, its radiance slot (16 B) written.  k_resolve: a
    16-B radiance slot read per path, the fp32 RGB accumulator read and written once per pixel and batch."""
    segs, hits, paths, lights = stats["segments"], stats["shaded"], stats["paths"], stats.get("ended_on_light", 0)
    misses = segs - hits
    queued = hits - lights              # hits that a shading launch sees
    scattered = segs - paths            # every non-camera segment was written by a shading launch
    ended_in_shade = queued - scattered  # other lights, exhausted depth budgets
    ended_in_trace = misses + lights
    # (camera rays that miss write no record and later enders also read 16 B of throughput: the two roughly cancel)
    return {"trace": (segs - paths) * 32 + paths * RAY_RECORD + queued * (HIT_RECORD + QUEUE_INDEX) + ended_in_trace * RADIANCE,
            "shade": queued * (QUEUE_INDEX + RAY_RECORD + HIT_RECORD) + scattered * RAY_RECORD + ended_in_shade * RADIANCE,
            "resolve": paths * RADIANCE}


def cpu_baseline(desc, npix, seed):
    """The oracle on every hardware thread of this host, ~10-30 s of CPU work on the same scene at full resolution."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle  # test infrastructure: used here only as the timed CPU baseline
    threads = max(1, min(1024, os.cpu_count() or 1))  # src/main.rs:57 spawns num_cpus::get() render tasks
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = threads
    oracle.render(desc, 1, seed=seed, threads=threads, pixel_stride=64)  # page in, start the threads once
    # whole passes run concurrently, one per thread, as the reference runs them (src/main.rs:56-63)
    passes = threads
    t0 = time.perf_counter()
    oracle.render(desc, passes, seed=seed, threads=threads)
    dt = time.perf_counter() - t0
    if dt < 1.0:  # a fast host: ~2 s of wall time on every thread
        passes = threads * max(2, min(16, int(round(2.0 / max(dt, 1e-3)))))
        t0 = time.perf_counter()
        oracle.render(desc, passes, seed=seed, threads=threads)
        dt = time.perf_counter() - t0
    # one thread, every 4th pixel of one pass (the rate does not depend on which pixels are sampled)
    t0 = time.perf_counter()
    oracle.render(desc, 1, seed=seed, threads=1, pixel_stride=4)
    one = (npix + 3) // 4 / (time.perf_counter() - t0) / 1e6
    return {"value": round(npix * passes / dt / 1e6, 4), "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "host_cpu_count": os.cpu_count(), "host_cpus_usable": usable, "value_1_thread": round(one, 4),
            "sample": f"{passes} full-resolution 1-spp pass(es) of the same scene, {threads} threads (= os.cpu_count()), "
                      f"reference-order (unordered candidate-set) BVH traversal"}


def c5_job(args, rank, world, local_rank, rehearse):
    """N > 1 only: the job the multi-GPU target is stated on -- data/zy_scene.json at 3840x2160, 4096 passes in total --
    split over the ranks by sample range (rank r renders passes r, r+N, ...: what the reference's concurrent tasks do on
    one host, src/main.rs:56-63) and summed onto rank 0 by one reduce: strong scaling.  Timed twice: once as a whole
    (passes and reduce queued back to back on the renderer's stream), once with a fence after the passes so that the
    render and the reduce times are known apart.  Before that, 64 passes rendered split and by rank 0 alone are compared."""
    import torch
    import torch.distributed as dist
    from v4ray_hip import Renderer, scenes
    from v4ray_hip.distributed import render_split
    w, h, _ = scenes.FULL_SIZE["c5"]
    total = args.c5_spp
    scene = scenes.load("c5", w, h, pretransform_triangles=True)
    r = Renderer(scene.renderer_param, scene.camera_param, scene, device=local_rank)
    dev = torch.device("cuda", local_rank)
    accum = torch.zeros(h, w, 3, dtype=torch.float32, device=dev)
    r.bind_accumulator(accum.data_ptr())
    stream = torch.cuda.ExternalStream(r.stream_ptr(), device=dev)

    def render(first, stride, n, rows=None):
        r.render_passes(n, first_pass=first, pass_stride=stride, rows=rows)
        if rehearse:
            r.synchronize()

    def fence():
        r.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def clear():
        with torch.cuda.stream(stream):
            accum.zero_()

    # the split changes nothing but the order of the final sum: 64 passes over all ranks against rank 0 alone
    check_passes = min(64, total)
    with torch.cuda.stream(stream):
        render_split(render, accum, check_passes, dst=0)
    fence()
    check = None
    if rank == 0:
        split_image = accum.clone()
        clear()
        render(0, 1, check_passes)
        r.synchronize()
        torch.cuda.synchronize()
        diff = (split_image - accum).abs().max().item()
        check = {"passes": check_passes, "mean_split": round(split_image.mean().item() / check_passes, 7),
                 "mean_single_rank": round(accum.mean().item() / check_passes, 7),
                 "max_abs_difference_of_sums": diff, "max_of_sums": round(accum.max().item(), 4)}
    fence()
    # whole job, nothing between the passes and the reduce
    clear()
    fence()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        render_split(render, accum, total, dst=0)
    fence()
    whole = time.perf_counter() - t0
    mean_job = accum.mean().item() / total if rank == 0 else 0.0
    # the same job with a fence after the passes: render and reduce apart
    clear()
    fence()
    t0 = time.perf_counter()
    first, stride, n = rank, world, (total - rank + world - 1) // world if total > rank else 0
    if n:
        render(first, stride, n)
    r.synchronize()
    mine = time.perf_counter() - t0
    fence()
    t1 = time.perf_counter()
    with torch.cuda.stream(stream):
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
    fence()
    reduce_s = time.perf_counter() - t1
    times = torch.tensor([whole, mine, reduce_s], dtype=torch.float64, device=dev)
    dist.all_reduce(times, op=dist.ReduceOp.MAX)
    whole, slowest, reduce_s = (float(x) for x in times.tolist())
    r.close()
    if rank != 0:
        return None
    return {"workload": "c5: " + WORKLOADS["c5"] + f", {w}x{h}, {total} passes in total", "scaling": "strong",
            "split": "sample-range (pass p on rank p mod N) + one reduce of the (H,W,3) fp32 sums onto rank 0",
            "backend": dist.get_backend(), "n_gpus": world, "passes_per_rank": (total + world - 1) // world,
            "value": round(w * h * total / whole / 1e6, 3), "unit": "Mpaths/s", "job_seconds": round(whole, 4),
            "render_seconds_slowest_rank": round(slowest, 4), "reduce_seconds": round(reduce_s, 5),
            "reduce_bytes": w * h * 3 * 4, "mean_radiance": round(mean_job, 7), "split_check": check}


def side_scene(Renderer, scenes, device, with_cpu):
    """The random-spheres scene (c2, 1080p) measured briefly beside the headline workload."""
    w, h, _ = scenes.FULL_SIZE["c2"]
    scene = scenes.load("c2", w, h)
    r = Renderer(scene.renderer_param, scene.camera_param, scene, device=device)
    spp, steps = 128, 2
    r.render_passes(spp)
    r.synchronize()
    r.clear()
    t0 = time.perf_counter()
    for k in range(steps):
        r.render_passes(spp, first_pass=k * spp)
    r.synchronize()
    dt = time.perf_counter() - t0
    value = w * h * spp * steps / dt / 1e6
    out = {"workload": "c2: " + WORKLOADS["c2"] + ", 1920x1080", "value": round(value, 3), "unit": "Mpaths/s",
           "spp_per_step": spp, "steps": steps}
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(r.desc, w * h, r.seed)
        out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    r.close()
    return out


def kernel_sources_sha256():
    """Hash of the device sources (the same files profiles/summarize_pmc_passes.py hashes when a counter summary is made)."""
    import hashlib
    h = hashlib.sha256()
    for rel in ("ray-tracing_amd/csrc/device/v4r_device.hip", "ray-tracing_amd/csrc/device/v4r_scene_dev.hpp",
                "ray-tracing_amd/csrc/device/v4r_math.hpp", "include/v4ray_hip.h"):
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


GROUP_KERNELS = {"trace": ("k_trace", "k_tail"), "shade": ("k_shade",), "resolve": ("k_resolve",)}
GROUP_TITLE = {"trace": "k_trace (+ k_tail): BVH traversal and intersection", "shade": "k_shade: material evaluation and scattering",
               "resolve": "k_resolve: per-pixel sum of the batch's path radiances"}


def roofline_block(r, args, npix, spp, ms_per_step):
    """Exclusive kernel times (HIP events, single stream) and exact work counts of PROFILE_PASSES more passes; the block
    prices the kernel group with the largest exclusive time on both of its possible bounds and names the nearer one."""
    n = min(PROFILE_PASSES, spp)
    r.enable_profile(2)
    r.read_profile_groups(reset=True)
    r.render_passes(n, first_pass=0)
    groups = r.read_profile_groups(reset=True)
    r.enable_profile(0)
    r.enable_stats(True)
    r.read_stats(reset=True)
    r.render_passes(n, first_pass=0)
    stats = r.read_stats(reset=True)
    r.enable_stats(False)
    paths = npix * n
    qb = queue_bytes(stats)
    # hardware counters of the same workload, collected by tools/profile_round.sh and committed under profiles/: they are
    # not measured in this run; they apply when the kernels are the ones that were profiled (same source hash)
    pmc, pmc_doc, pmc_path = None, None, PMC_SUMMARY.get(args.workload)
    if pmc_path and os.path.exists(os.path.join(ROOT, pmc_path)) and not (args.width or args.height) and \
            (args.workload != "c4" or args.c4_subdivide == 2) and not args.flags:
        with open(os.path.join(ROOT, pmc_path)) as f:
            pmc_doc = json.load(f)
        pmc = pmc_doc["kernels"]
    pmc_current = bool(pmc_doc) and pmc_doc.get("kernel_sources_sha256") == kernel_sources_sha256()

    def pmc_rows(prefixes):
        return [v for k, v in pmc.items() if k.startswith(tuple(prefixes))] if pmc else []

    per_group = {}
    for g, kernels in GROUP_KERNELS.items():
        ms, launches = groups[g]["ms"], max(1, groups[g]["launches"])
        secs = ms * 1e-3
        e = {"kernels": list(kernels), "launches": launches, "avg_launch_us": round(ms * 1e3 / launches, 3),
             "ms_per_64_passes": round(ms, 3),
             "algorithmic_bytes_per_launch": round(qb[g] / launches),
             "algorithmic_bytes_per_path": round(qb[g] / paths, 1),
             "achieved_GBps": round(qb[g] / secs / 1e9, 1) if ms else None}
        e["frac"] = round(e["achieved_GBps"] / HBM_PEAK_GBS, 4) if ms else None
        rows = pmc_rows(kernels)
        if rows and ms:
            per_path = sum(v["hbm_bytes_per_path"] for v in rows)
            e["traffic_bytes_per_path"] = round(per_path, 1)
            e["traffic_bytes_per_launch"] = round(per_path * paths / launches)
            e["traffic_GBps"] = round(per_path * paths / secs / 1e9, 1)
            e["traffic_frac"] = round(e["traffic_GBps"] / HBM_PEAK_GBS, 4)
            valu = sum(v.get("valu_insts", 0.0) for v in rows)
            if valu:
                per_path_insts = valu / pmc_doc["paths"]
                lanes = sum(v["valu_insts"] * v["active_lanes_per_valu_inst"] for v in rows if v.get("valu_insts")) / valu
                rate, peak = per_path_insts * paths / secs, SIMDS * PEAK_CLOCK_HZ / VALU_ISSUE_CYCLES
                e["issue"] = {"valu_wave_insts_per_path": round(per_path_insts, 2), "achieved_Ginst_per_s": round(rate / 1e9, 2),
                              "peak_Ginst_per_s": round(peak / 1e9, 1), "frac": round(rate / peak, 4),
                              "active_lanes_per_valu_inst": round(lanes, 1), "lane_frac": round(lanes / 64, 3)}
        per_group[g] = e
    total_bytes = sum(qb.values()) + npix * 24 * max(1, n // 32)
    # every queue byte of a full step over the step's wall time must stay under the peak
    step_bytes = total_bytes * spp / n
    whole = {"algorithmic_bytes_per_step": round(step_bytes), "ms_per_step": ms_per_step,
             "GBps": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
             "frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    total_ms = max(1e-9, sum(g["ms"] for g in groups.values()))
    dominant = max(groups, key=lambda g: groups[g]["ms"])
    d = per_group[dominant]
    hbm_frac = d.get("traffic_frac") if d.get("traffic_frac") is not None else d["frac"]
    issue = d.get("issue")
    out = {"kernel": GROUP_TITLE[dominant], "share_of_device_time": round(groups[dominant]["ms"] / total_ms, 3)}
    if issue and issue["frac"] > (hbm_frac or 0.0):
        out.update({"bound": "issue", "achieved": issue["achieved_Ginst_per_s"], "peak": issue["peak_Ginst_per_s"],
                    "unit": "G wave64 vector instructions/s", "frac": issue["frac"]})
    else:
        out.update({"bound": "hbm", "achieved": d.get("traffic_GBps") or d["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac})
    tables = {"note": "BVH nodes and primitive records read per ray, served by LDS / L1 / L2, not HBM",
              "node_bytes_per_path": round(stats["nodes"] * NODE / paths, 1),
              "prim_bytes_per_path": round(stats["prims"] * PRIM / paths, 1),
              "texel_bytes_per_path": round(stats["texel_fetches"] * TEXEL / paths, 2)}
    out.update({
        "traffic": d.get("traffic_bytes_per_launch"),
        # both bounds of the dominant group, whichever was named above
        "hbm_algorithmic_frac": d["frac"], "hbm_traffic_frac": d.get("traffic_frac"), "issue_frac": issue["frac"] if issue else None,
        "algorithmic_bytes_per_path": d["algorithmic_bytes_per_path"],
        "table_bytes_per_path": round(tables["node_bytes_per_path"] + tables["prim_bytes_per_path"], 1) if dominant == "trace" else None,
        "avg_launch_us": d["avg_launch_us"],
        "pmc": None if not pmc_doc else {
            "source": pmc_path, "measured_in_this_run": False,
            "kernel_sources_match": pmc_current,
            "note": "FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU / SQ_THREAD_CYCLES_VALU per path of the same workload, collected by rocprofv3 "
                    "in separate passes and committed; scaled by this run's paths and measured kernel time"
                    + ("" if pmc_current else " -- the kernel sources have changed since: traffic and issue figures are stale")},
        "note": "hbm_algorithmic_frac: queue records the group must move / its exclusive time / 8 TB/s; hbm_traffic_frac: HBM bytes the "
                "counters saw; issue_frac: wave64 vector instructions x 2 cycles / SIMD cycles.  `bound` is the larger of traffic and issue",
        "groups": per_group, "whole_step_vs_wall": whole, "tables": tables})
    out["work"] = {"profiled_passes": n,
                   "nodes_per_segment": round(stats["nodes"] / stats["segments"], 3),
                   "prims_per_segment": round(stats["prims"] / stats["segments"], 3),
                   "segments_per_path": round(stats["segments"] / stats["paths"], 3),
                   "lane_utilisation_node_step": round(stats["nodes"] / max(1, 64 * stats["wave_node_steps"]), 3),
                   "lane_utilisation_prim_test": round(stats["prims"] / max(1, 64 * stats["wave_leaf_steps"]), 3),
                   "wave_node_steps_per_wave_ray": round(64 * stats["wave_node_steps"] / stats["segments"], 2),
                   "wave_prim_tests_per_wave_ray": round(64 * stats["wave_leaf_steps"] / stats["segments"], 2),
                   "queue_bytes_per_path": {g: round(b / paths, 1) for g, b in qb.items()}}
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")

    import torch
    import torch.distributed as dist
    from v4ray_hip import Renderer, scenes
    from v4ray_hip.distributed import render_split

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # V4R_BENCH_REHEARSE=1: every rank on GPU 0 with the gloo backend, to rehearse the N > 1 code path on a
    # one-GPU box (RCCL refuses two ranks on one device); the numbers of such a run mean nothing.
    rehearse = os.environ.get("V4R_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    fw, fh, fspp = scenes.FULL_SIZE[args.workload]
    width, height = args.width or fw, args.height or fh
    spp = args.spp or fspp
    # the bunny's mesh transform is baked into its triangles on the host: the same geometry without a per-test
    # ray transform (tests/test_gpu_parity.py::test_primary_hits_pretransformed_bunny checks ids and t)
    extra = {"subdivide": args.c4_subdivide} if args.workload == "c4" and args.c4_subdivide != 2 else {}
    scene = scenes.load(args.workload, width, height, pretransform_triangles=True, flags=args.flags, **extra)
    r = Renderer(scene.renderer_param, scene.camera_param, scene, device=local_rank)
    npix = width * height
    accum = torch.zeros(height, width, 3, dtype=torch.float32, device=f"cuda:{local_rank}")
    r.bind_accumulator(accum.data_ptr())
    # Everything that touches the accumulator (clears, the collective) is queued on the renderer's own stream, so it
    # is ordered against the passes without blocking the host: the reduce of one step can still be in flight while
    # the host queues the next step's launches behind it.
    stream = torch.cuda.ExternalStream(r.stream_ptr(), device=torch.device("cuda", local_rank))

    def render(first, stride, n, rows=None):
        r.render_passes(n, first_pass=first, pass_stride=stride, rows=rows)
        if world > 1 and rehearse:
            r.synchronize()  # gloo stages through the host and knows nothing about streams

    def step():
        # rank r: passes r, r+N, ... of an (N * spp)-pass job, then one RCCL reduce of the accumulators
        with torch.cuda.stream(stream):
            render_split(render, accum, spp * world, dst=0, split=args.split)

    def clear():
        with torch.cuda.stream(stream):
            accum.zero_()

    def fence():
        r.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if world > 1:
        # the communicator's lazy set-up (connections, buffers for a message of this size) is not part of any step, whatever --warmup is
        with torch.cuda.stream(stream):
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        fence()
    for _ in range(args.warmup):
        clear()
        step()
    fence()
    clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = round(elapsed / args.steps * 1e3, 3)

    roofline = None
    if rank == 0 and not args.no_profile:
        roofline = roofline_block(r, args, npix, spp, ms_per_step)
    if world > 1:
        dist.barrier()

    if rank == 0:
        paths = npix * spp * world * args.steps
        out = {
            "metric": "Mpaths/s (pixels x spp / s)", "value": round(paths / elapsed / 1e6, 3), "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: " + (WORKLOADS[args.workload] if args.workload != "c4" or args.c4_subdivide == 2
                                                          else f"bunny.obj, {args.c4_subdivide} subdivision(s) + earth-textured sphere in a Cornell box"),
                       "width": width, "height": height, "spp_per_gpu_per_step": spp, "max_depth": scene.renderer_param.max_depth,
                       "paths_per_step": npix * spp * world,
                       "split": ("single GPU" if world == 1 else
                                 ("sample-range (pass p on rank p mod N)" if args.split == "sample" else "tile (rank r renders its band of rows)")
                                 + (" + RCCL reduce" if dist.get_backend() == "nccl" else f" + {dist.get_backend()} reduce (rehearsal)"))},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(r.desc, npix, r.seed)
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    r.close()
    if world > 1 and args.c5_spp > 0:
        job = c5_job(args, rank, world, local_rank, rehearse)
        if rank == 0:
            out["c5_job"] = job
    if rank == 0:
        if world == 1 and args.workload == "c4" and not args.no_side and not (args.width or args.height):
            out["spheres_scene"] = side_scene(Renderer, scenes, local_rank, with_cpu=not args.no_cpu_baseline)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
