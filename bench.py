#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s of the render-loop hot path on N MI355X GPUs of one node.

Workload (config.workload): by default the Book-1 random-spheres scene (c2) at
1920x1080, 256 samples per pixel per GPU.  One "step" renders those 256 full-image 1-spp passes on every
rank (rank r takes passes r, r+N, r+2N, ... of one N*256-spp job, so per-GPU work is fixed: weak scaling)
and, for N > 1, sums the per-GPU radiance accumulators onto rank 0 with one RCCL reduce over xGMI.
Scene tables are resident in HBM before the timed region starts; nothing crosses PCIe inside it.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (BVH traversal + intersection,
k_trace) in algorithmic bytes per second against the 8 TB/s HBM peak: its device time comes from HIP
events recorded around every k_trace launch inside the timed region, the bytes from the work counters of
an instrumented (untimed) run of the same passes.  `cpu_baseline` is the CPU oracle (the scalar C++
restatement of the reference's loop, oracle/) timed on this host's cores on a bounded sample of the same
workload -- a reported baseline, not the optimisation target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
# Bytes of the traversal kernel per unit of work (the record sizes of its queues and tables):
RAY_RECORD, HIT_RECORD, NODE, PRIM, RADIANCE = 48, 20, 128, 48, 16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--spp", type=int, default=0, help="passes per GPU per step (default: the workload's full count)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record HIP events around k_trace")
    ap.add_argument("--split", default="sample", choices=["sample", "tile"],
                    help="N > 1: rank r renders passes r, r+N, ... of the job (sample) or all passes of its band of rows (tile)")
    ap.add_argument("--c4-subdivide", type=int, default=2, help="midpoint subdivisions of bunny.obj in c4 (2: 79 488 triangles, 0: 4 968)")
    ap.add_argument("--no-bunny", action="store_true", help="skip the short bunny-scene (c4) measurement added to the c2 line")
    return ap.parse_args()


def trace_bytes(stats):
    """Algorithmic bytes of k_trace for the counted work: one 48-B ray record per segment (read from the queue, or for a
    camera ray that hits something written to it), 128 B per 4-wide BVH node and 48 B per primitive fetched, per hit a
    16-B hit record + a 4-B class-queue index, per miss one 16-B radiance write."""
    hits = stats["shaded"]
    misses = stats["segments"] - hits
    return (stats["segments"] * RAY_RECORD + stats["nodes"] * NODE + stats["prims"] * PRIM + hits * HIT_RECORD
            + misses * RADIANCE)


def cpu_baseline(desc, npix, seed):
    """The oracle on this host's cores, ~10-30 s of CPU work on the same scene at full resolution."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle  # test infrastructure: used here only as the timed CPU baseline
    cores = max(1, min(16, os.cpu_count() or 1))
    oracle.render(desc, 1, seed=seed, threads=cores, pixel_stride=64)  # page in
    t0 = time.perf_counter()
    oracle.render(desc, 1, seed=seed, threads=cores)
    dt = time.perf_counter() - t0
    passes = max(1, min(16, int(round(1.5 / max(dt, 1e-3)))))  # ~1.5 s of wall time on every core
    if passes > 1:
        t0 = time.perf_counter()
        oracle.render(desc, passes, seed=seed, threads=cores)
        dt = time.perf_counter() - t0
    # one thread, every 4th pixel of one pass (the rate does not depend on which pixels are sampled)
    t0 = time.perf_counter()
    oracle.render(desc, 1, seed=seed, threads=1, pixel_stride=4)
    one = (npix + 3) // 4 / (time.perf_counter() - t0) / 1e6
    return {"value": round(npix * passes / dt / 1e6, 4), "unit": "Mpaths/s", "cores": cores, "kind": "port", "value_1_thread": round(one, 4),
            "sample": f"{passes} full-resolution 1-spp pass(es) of the same scene, {cores} threads, "
                      f"reference-order (unordered candidate-set) BVH traversal"}


def bunny_scene(Renderer, scenes, device):
    """The scene the >= 100x GPU/CPU target is stated on (c4: 79 488-triangle bunny, 1080p), measured briefly
    beside the headline workload: 2 x 128 passes on the GPU, the oracle on the host cores."""
    w, h, _ = scenes.FULL_SIZE["c4"]
    scene = scenes.load("c4", w, h, pretransform_triangles=True)
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
    cpu = cpu_baseline(r.desc, w * h, r.seed)
    r.close()
    return {"workload": "c4: bunny.obj subdivided to 79 488 triangles + earth-textured sphere in a Cornell box, 1920x1080",
            "value": round(value, 3), "unit": "Mpaths/s", "spp_per_step": spp, "steps": steps,
            "cpu_baseline": cpu, "gpu_over_cpu": round(value / cpu["value"], 1)}


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
    scene = scenes.load(args.workload, width, height, pretransform_triangles=True, **extra)
    r = Renderer(scene.renderer_param, scene.camera_param, scene, device=local_rank)
    npix = width * height
    accum = torch.zeros(height, width, 3, dtype=torch.float32, device=f"cuda:{local_rank}")
    r.bind_accumulator(accum.data_ptr())

    def render(first, stride, n, rows=None):
        r.render_passes(n, first_pass=first, pass_stride=stride, rows=rows)
        if world > 1:
            r.synchronize()  # the renderer runs on its own stream; the collective runs on torch's

    def step():
        # rank r: passes r, r+N, ... of an (N * spp)-pass job, then one RCCL reduce of the accumulators
        render_split(render, accum, spp * world, dst=0, split=args.split)

    def fence():
        r.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        accum.zero_()
        step()
    fence()
    profile = not args.no_profile
    if profile:
        r.enable_profile(True)
        r.read_profile(reset=True)
    accum.zero_()
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
    trace_ms, trace_launches = r.read_profile(reset=True) if profile else (0.0, 0)
    r.enable_profile(False)

    roofline = None
    stats = None
    if rank == 0 and profile and trace_launches:
        # the same passes once more with the instrumented kernels (untimed): exact work counts of one step
        r.enable_stats(True)
        r.read_stats(reset=True)
        r.render_passes(spp, first_pass=rank, pass_stride=world)
        stats = r.read_stats(reset=True)
        r.enable_stats(False)
        bytes_per_step = trace_bytes(stats)
        achieved = bytes_per_step * args.steps / (trace_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_trace", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "note": "algorithmic bytes are mostly BVH nodes and primitives served by L1/L2 (tables <= 15 MB), so "
                            "achieved can exceed the HBM peak; traffic = HBM bytes per launch from rocprofv3 PMC",
                    "avg_launch_us": round(trace_ms * 1e3 / trace_launches, 3), "launches": trace_launches,
                    "kernel_ms_per_step": round(trace_ms / args.steps, 3),
                    "algorithmic_bytes_per_path": round(bytes_per_step / (npix * spp), 2),
                    "nodes_per_segment": round(stats["nodes"] / stats["segments"], 3),
                    "prims_per_segment": round(stats["prims"] / stats["segments"], 3),
                    "segments_per_path": round(stats["segments"] / stats["paths"], 3),
                    "lane_utilisation_node_step": round(stats["nodes"] / max(1, 64 * stats["wave_node_steps"]), 3),
                    "lane_utilisation_prim_test": round(stats["prims"] / max(1, 64 * stats["wave_leaf_steps"]), 3),
                    "wave_node_steps_per_wave_ray": round(64 * stats["wave_node_steps"] / stats["segments"], 2),
                    "wave_prim_tests_per_wave_ray": round(64 * stats["wave_leaf_steps"] / stats["segments"], 2)}
        # HBM bytes of k_trace per launch: FETCH_SIZE / WRITE_SIZE cannot be read from inside this process, so
        # the per-path figure of the committed rocprofv3 --pmc passes over this workload is scaled to this run.
        pmc = os.path.join(ROOT, "profiles", f"r01_{args.workload}_pmc_traffic.json")
        if os.path.exists(pmc) and not (args.width or args.height):
            with open(pmc) as f:
                per_path = json.load(f)["kernels"]["k_trace"]["hbm_bytes_per_path"]
            roofline["traffic"] = round(per_path * npix * spp * args.steps / trace_launches)
            roofline["algorithmic_bytes_per_launch"] = round(bytes_per_step * args.steps / trace_launches)
            roofline["traffic_source"] = os.path.relpath(pmc, ROOT)
    if world > 1:
        dist.barrier()

    if rank == 0:
        paths = npix * spp * world * args.steps
        out = {
            "metric": "Mpaths/s (pixels x spp / s)", "value": round(paths / elapsed / 1e6, 3), "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: " + {
                "c1": "data/scene.json verbatim (Cornell walls, 4 968-triangle bunny, fog sphere, 4 985 shapes)",
                "c2": "Book-1 random-spheres scene (484 spheres, Lambertian/metal/dielectric)",
                "c3": "Cornell box, cosine + light-list mixture sampling",
                "c4": ("bunny.obj subdivided to 79 488 triangles" if args.c4_subdivide == 2 else f"bunny.obj, {args.c4_subdivide} subdivision(s)")
                      + " + earth-textured sphere in a Cornell box",
                "c5": "zy_scene.json (noise/image-textured spheres in a Cornell box)"}[args.workload],
                "width": width, "height": height, "spp_per_gpu_per_step": spp, "max_depth": scene.renderer_param.max_depth,
                "paths_per_step": npix * spp * world,
                "split": ("single GPU" if world == 1 else "sample-range (pass p on rank p mod N) + RCCL reduce" if args.split == "sample"
                          else "tile (rank r renders its band of rows) + RCCL reduce")},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(r.desc, npix, r.seed)
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    r.close()
    if rank == 0:
        if world == 1 and args.workload == "c2" and not args.no_bunny and not args.no_cpu_baseline and not (args.width or args.height):
            out["bunny_scene"] = bunny_scene(Renderer, scenes, local_rank)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
