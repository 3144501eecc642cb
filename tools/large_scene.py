#!/usr/bin/env python3
"""Larger meshes than any quoted workload (GPU box): the c4 scene with bunny.obj subdivided 3, 4 and 5 times (318 k, 1.27 M,
5.09 M triangles) at 1920x1080 -- build time, primary hits against the oracle, and the rate of a 128-pass job."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from v4ray_hip import Renderer, scenes  # noqa: E402
import oracle  # noqa: E402

MISS = 0xFFFFFFFF
for sub in [int(a) for a in sys.argv[1:]] or [3, 4, 5]:
    t0 = time.time()
    scene = scenes.load("c4", 1920, 1080, subdivide=sub)
    t1 = time.time()
    r = Renderer(scene.renderer_param, scene.camera_param, scene)
    t2 = time.time()
    prim_d, t_d = r.trace_primary()
    prim_o, t_o = oracle.trace_primary(r.desc)
    t3 = time.time()
    same = prim_d == prim_o
    hit = same & (prim_o != MISS)
    rel = np.zeros(prim_o.shape)
    rel[hit] = np.abs(t_d[hit].astype(np.float64) - t_o[hit]) / np.abs(t_o[hit])
    r.render_passes(128)
    r.synchronize()
    r.clear()
    t4 = time.time()
    r.render_passes(128)
    r.synchronize()
    t5 = time.time()
    tree = r.read_tree()
    print(json.dumps({"subdivide": sub, "n_prims": int(r.desc.contents.n_prims), "tree_width": tree["width"], "n_nodes": int(tree["nodes"].shape[0]),
                      "host_scene_s": round(t1 - t0, 2), "renderer_create_s": round(t2 - t1, 2), "oracle_and_device_primary_s": round(t3 - t2, 2),
                      "primary_hits": float((prim_o != MISS).mean()), "triangles_seen": int(np.unique(prim_o).size), "primary_ids_equal": float(same.mean()), "primary_ok_1e-5": float((same & (rel <= 1e-5)).mean()), "max_rel_t": float(rel.max()),
                      "Mpaths_per_s_128spp": round(1920 * 1080 * 128 / (t5 - t4) / 1e6, 1)}), flush=True)
    r.close()
