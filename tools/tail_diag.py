#!/usr/bin/env python3
"""Work counters of one workload for several tail-start segments (GPU diagnostic)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing_amd"))
from v4ray_hip import Renderer, scenes  # noqa: E402

name = sys.argv[1]
w, h = int(sys.argv[2]), int(sys.argv[3])
for give_up in os.environ.get("DIAG_GIVE_UP", "8,0").split(","):
    for start in sys.argv[4:]:
        os.environ["V4R_TAIL_SEGMENT"] = start
        os.environ["V4R_GIVE_UP_LANES"] = give_up
        s = scenes.load(name, w, h)
        r = Renderer(s.renderer_param, s.camera_param, s)
        r.enable_stats(True)
        r.render_passes(3)
        img, _ = r.read_accumulator()
        st = r.read_stats()
        print(f"give_up {give_up} tail {start:>4}: mean {float(img.mean()):.6f}", {k: st[k] for k in ("segments", "paths", "shaded", "nodes", "prims")}, flush=True)
        r.close()
