"""The host library's C API under AddressSanitizer + UndefinedBehaviorSanitizer (host code only): tools/host_sanitize.cpp
is built together with the host sources and run over the reference's scene files, the image readers / writers and the
checkpoint files, damaged copies included."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_api_under_sanitizers(tmp_path):
    sources = [s for s in glob.glob(os.path.join(ROOT, "ray-tracing_amd", "csrc", "host", "*.cpp")) if not s.endswith("main.cpp")]
    exe = tmp_path / "host_sanitize"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "ray-tracing_amd", "csrc", "host"),
           os.path.join(ROOT, "tools", "host_sanitize.cpp"), *sources, "-o", str(exe), "-lpthread"]
    built = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    if built.returncode != 0 and "asan" in built.stdout.lower() and "cannot find" in built.stdout.lower():
        pytest.skip("the sanitizer runtimes are not installed")
    assert built.returncode == 0, built.stdout[-3000:]
    work = tmp_path / "out"
    work.mkdir()
    run = subprocess.run([str(exe), os.path.join(ROOT, "assets"), str(work)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", LD_PRELOAD=os.environ.get("LD_PRELOAD", "")))
    assert run.returncode == 0 and "host_sanitize ok" in run.stdout, run.stdout[-4000:]
