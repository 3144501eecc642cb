"""Print the surface-area model's pricing of the 4-wide and the 8-wide tree for each configured scene (needs a GPU: the
renderer is created for real).  The library prints the line itself under V4R_VERBOSE=1."""
import os, sys
os.environ["V4R_VERBOSE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "ray-tracing_amd"))
from v4ray_hip import Renderer, scenes
for w in sys.argv[1:] or ["c1", "c2", "c3", "c4", "c5"]:
    fw, fh, _ = scenes.FULL_SIZE[w]
    scene = scenes.load(w, fw, fh, pretransform_triangles=True)
    print(w, flush=True)
    Renderer(scene.renderer_param, scene.camera_param, scene, device=0).close()
