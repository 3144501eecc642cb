#!/usr/bin/env python3
"""How much of a job's time goes to the long, nearly empty tail of segments: the same workload rendered with
smaller depth budgets (GPU).  usage: depth_sweep.py <workload> <spp> <depth>..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing_amd"))
from v4ray_hip import Renderer, RendererParam, scenes  # noqa: E402

name, spp = sys.argv[1], int(sys.argv[2])
for depth in [int(a) for a in sys.argv[3:]]:
    s = scenes.load(name, pretransform_triangles=True)
    p = s.renderer_param
    r = Renderer(RendererParam(p.width, p.height, depth, p.antialias), s.camera_param, s)
    r.render_passes(spp)
    r.synchronize()
    best = 1e9
    for k in range(3):
        t0 = time.perf_counter()
        r.render_passes(spp, first_pass=(k + 1) * spp)
        r.synchronize()
        best = min(best, time.perf_counter() - t0)
    img, n = r.read_accumulator()
    print(f"{name} depth {depth:3d}: {p.width * p.height * spp / best / 1e6:9.1f} Mpaths/s  mean {float(img.mean()) / n:.6f}", flush=True)
    r.close()
