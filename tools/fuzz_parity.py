#!/usr/bin/env python3
"""Parity fuzzer (GPU): random mixed scenes (tests/test_gpu_parity.py::random_scene) x random rays, device closest hit
against the oracle's reference-order traversal.  Prints the worst agreement; exits non-zero below the thresholds.

    python tools/fuzz_parity.py [first_seed] [n_scenes] [n_shapes]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing_amd"))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
from test_gpu_parity import random_scene  # noqa: E402
from v4ray_hip import PerspectiveCameraParam, Renderer, RendererParam  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shapes = int(sys.argv[3]) if len(sys.argv) > 3 else 300
worst_same, worst_rel, worst_mean = 1.0, 0.0, 0.0
for seed in range(first, first + count):
    scene = random_scene(seed, shapes)
    rng = np.random.default_rng(1000 + seed)
    # random camera: position, field of view, lens aperture / focus, antialiasing on or off, odd image sizes
    eye = rng.normal(size=3)
    eye = tuple(float(x) for x in eye / np.linalg.norm(eye) * rng.uniform(5, 40))
    cam = PerspectiveCameraParam(eye, tuple(float(x) for x in rng.uniform(-3, 3, size=3)), float(rng.uniform(20, 90)),
                                 aperture=float(rng.choice([0.0, 0.3, 2.0])), focus_dist=float(rng.uniform(5, 30)))
    param = RendererParam(int(rng.integers(17, 97)), int(rng.integers(9, 71)), max_depth=int(rng.integers(1, 30)),
                          antialias=bool(rng.integers(0, 2)))
    r = Renderer(param, cam, scene)
    n = 30000
    o = rng.uniform(-14, 14, size=(n, 3)).astype(np.float32)
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    pd, td = r.trace_rays(o, v, seed=seed)
    po, to = oracle.trace_rays(r.desc, o, v, seed=seed)
    same = pd == po
    hit = same & (po != 0xFFFFFFFF)
    rel = np.abs(td[hit] - to[hit]) / np.abs(to[hit])
    r.render_passes(4)
    img_d, _ = r.read_accumulator()
    img_o, _ = oracle.render(r.desc, 4, seed=r.seed)
    dm = abs(np.nanmean(img_d) - np.nanmean(img_o)) / (abs(np.nanmean(img_o)) + 1e-9)
    ok = np.isfinite(img_o).all(axis=2) & np.isfinite(img_d).all(axis=2)
    scale = np.abs(img_o).max(axis=2) + 0.04
    close = float(((np.abs(img_d - img_o).max(axis=2) / scale)[ok] < 2e-3).mean())
    print(f"seed {seed}: hit {hit.mean():.3f} image mean {np.nanmean(img_d):.5f} bit-identical pixels {float((img_d == img_o).all(axis=2).mean()):.4f}  ids {same.mean():.5f}  rel t p99.9 {np.quantile(rel, 0.999):.2e}  image mean diff {dm:.2e}  "
          f"pixels within 2e-3 {close:.4f}  non-finite pixels dev/oracle {int((~np.isfinite(img_d)).any(axis=2).sum())}/"
          f"{int((~np.isfinite(img_o)).any(axis=2).sum())}", flush=True)
    worst_same, worst_rel, worst_mean = min(worst_same, same.mean()), max(worst_rel, np.quantile(rel, 0.999)), max(worst_mean, dm)
    r.close()
print(f"worst: ids {worst_same:.5f}, rel t {worst_rel:.2e}, image mean {worst_mean:.2e}")
sys.exit(0 if worst_same >= 0.999 and worst_rel <= 1e-5 and worst_mean <= 0.02 else 1)
