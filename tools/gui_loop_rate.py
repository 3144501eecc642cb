"""Passes per second of the reference-style front end: N coroutines that each `await renderer.render()` and add the result
to a RenderResult, forever (main.py:1986-1990), against the same loop with one call at a time.
usage: python tools/gui_loop_rate.py [workload] [seconds]"""
import asyncio
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracing_amd"))
from v4ray_hip import Renderer, RenderResult, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
w, h, _ = scenes.FULL_SIZE[name]
scene = scenes.load(name, w, h, pretransform_triangles=True)
r = Renderer(scene.renderer_param, scene.camera_param, scene)
r.render_pass()


async def run(n_tasks, shared):
    result = RenderResult(w, h)
    stop = time.perf_counter() + seconds

    async def worker():
        while time.perf_counter() < stop:
            result.add(await r.render(shared=shared))
    t0 = time.perf_counter()
    await asyncio.gather(*[worker() for _ in range(n_tasks)])
    return result._count / (time.perf_counter() - t0)

for shared in (False, True):
    for n in (1, 4, 16, 64):
        rate = asyncio.run(run(n, shared))
        print(f"{name} {w}x{h}: {n:3d} render({'shared=True' if shared else ''}) in flight: {rate:7.1f} passes/s = {rate * w * h / 1e6:7.1f} Mpaths/s", flush=True)
