"""The checked build of the device library (lib/libv4ray_hip_checked.so, -DV4R_DEBUG_CHECKS): every traversal-stack push
and pop, queue index, queue length, ray-counter index and path slot is checked on the device, and a render call returns an
error if any check failed.  Rendering the configured scenes and a random 2000-shape scene with it must raise nothing."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CHECKED = os.path.join(os.path.dirname(HERE), "ray-tracing_amd", "lib", "libv4ray_hip_checked.so")


@pytest.mark.parametrize("tree", ["chosen", "4", "8"])
def test_no_bounds_check_fails(tree):
    """With the tree the renderer chooses per scene, and with either shape forced (the 8-wide tree pushes up to seven
    entries per node step)."""
    assert os.path.exists(CHECKED), "build it with make -C ray-tracing_amd checked"
    env = dict(os.environ, V4R_DEVICE_LIB=CHECKED)
    if tree != "chosen":
        env["V4R_WIDE"] = tree
    proc = subprocess.run([sys.executable, os.path.join(HERE, "checked_build_child.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, timeout=280)
    assert proc.returncode == 0, proc.stdout[-3000:]
    assert "random scene ok" in proc.stdout


def test_a_failed_check_is_reported():
    """Told a smaller queue capacity than the scene needs (V4R_DEBUG_CAP, checked build only), the kernels refuse the
    queue and the render call returns the error."""
    code = ("import sys; sys.path.insert(0, %r); from v4ray_hip import Renderer, scenes, V4RayError\n"
            "s = scenes.load('c3', 64, 64); r = Renderer(s.renderer_param, s.camera_param, s)\n"
            "try:\n    r.render_passes(2)\nexcept V4RayError as e:\n    print('reported:', e); sys.exit(7)\nsys.exit(0)\n"
            % os.path.join(os.path.dirname(HERE), "ray-tracing_amd"))
    env = dict(os.environ, V4R_DEVICE_LIB=CHECKED, V4R_DEBUG_CAP="100")
    proc = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert proc.returncode == 7 and "queue length" in proc.stdout, proc.stdout[-2000:]
