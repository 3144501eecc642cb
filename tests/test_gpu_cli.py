"""GPU test of the command-line renderer (bin/v4ray-hip): reference flags, progressive log lines, image output
and checkpoint/resume."""
import json
import os
import subprocess

import numpy as np
import pytest

from v4ray_hip import load_checkpoint, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "ray-tracing_amd", "bin", "v4ray-hip")


def run(args, cwd):
    p = subprocess.run([CLI] + args, cwd=cwd, capture_output=True, text=True, timeout=100)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def read_bmp(path):
    raw = open(path, "rb").read()
    assert raw[:2] == b"BM"
    w, h = int.from_bytes(raw[18:22], "little"), int.from_bytes(raw[22:26], "little")
    row = (w * 3 + 3) & ~3
    px = np.frombuffer(raw, dtype=np.uint8, offset=54).reshape(h, row)[:, :w * 3].reshape(h, w, 3)
    return px[::-1, :, ::-1]


def test_cli_renders_reference_scene_and_resumes(tmp_path):
    if not os.path.exists(CLI):
        pytest.fail(f"{CLI} is missing: build it with `make -C ray-tracing_amd cli`")
    # data/scene.json with a small image; run from the assets directory so bunny.obj / earthmap.jpg resolve
    with open(os.path.join(scenes.ASSETS, "scene.json")) as f:
        doc = json.load(f)
    doc["renderer"]["width"], doc["renderer"]["height"] = 96, 96
    scene_file = os.path.join(scenes.ASSETS, "_cli_test_scene.json")
    with open(scene_file, "w") as f:
        json.dump(doc, f)
    try:
        one = str(tmp_path / "one.bmp")
        out = run(["-i", scene_file, "-o", one, "--spp", "8", "--batch", "4"], scenes.ASSETS)
        assert "Iter 4 +" in out and "Iter 8 +" in out and "Iter 8 saved" in out   # src/main.rs:48-52,74
        img = read_bmp(one)
        assert img.shape == (96, 96, 3) and img.std() > 10
        two, ck = str(tmp_path / "two.bmp"), str(tmp_path / "job.ckpt")
        run(["-i", scene_file, "-o", two, "--spp", "4", "--batch", "4", "--checkpoint", ck], scenes.ASSETS)
        assert load_checkpoint(ck)[1] == 4
        out = run(["-i", scene_file, "-o", two, "--spp", "8", "--batch", "4", "--checkpoint", ck], scenes.ASSETS)
        assert "Resumed at iter 4" in out and "Iter 8 saved" in out
        assert load_checkpoint(ck)[1] == 8
        assert np.array_equal(read_bmp(two), img)
        # two renderers from one process (here both on GPU 0): passes split k, k+2, ..., sums merged on the host
        multi = str(tmp_path / "multi.bmp")
        out = run(["-i", scene_file, "-o", multi, "--spp", "8", "--batch", "4", "--devices", "0,0"], scenes.ASSETS)
        assert "Iter 8 saved" in out
        diff = np.abs(read_bmp(multi).astype(int) - img.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 0.01      # fp32 sums in another order, then 8-bit tonemap
    finally:
        os.remove(scene_file)
