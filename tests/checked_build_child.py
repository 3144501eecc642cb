"""Renders every configured scene and a random mixed scene with the library this process was pointed at (V4R_DEVICE_LIB);
tests/test_gpu_checked_build.py runs it with the checked build, whose entry points fail when a device-side bounds check did."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from v4ray_hip import PerspectiveCameraParam, Renderer, RendererParam, scenes  # noqa: E402

for name, size in (("c1", (160, 160)), ("c2", (256, 144)), ("c3", (160, 160)), ("c4", (256, 144)), ("c5", (256, 144)), ("c4", (640, 360))):
    scene = scenes.load(name, *size)
    r = Renderer(scene.renderer_param, scene.camera_param, scene)
    r.enable_stats(name == "c4")          # the instrumented kernels too
    r.render_passes(40)
    img, n = r.read_accumulator()
    assert n == 40 and np.isfinite(img).all()
    if size[0] * size[1] < 100000:
        r.render_passes_noalloc(3, first_pass=40)
    r.trace_primary()
    r.close()
    print(name, size, "ok", float(img.mean() / 40), flush=True)
from test_gpu_parity import random_scene  # noqa: E402
r = Renderer(RendererParam(96, 64, max_depth=12), PerspectiveCameraParam((0, 0, -30), (0, 0, 0), 50.0), random_scene(4, 2000))
r.render_passes(16)
r.read_accumulator()
r.close()
print("random scene ok", flush=True)
