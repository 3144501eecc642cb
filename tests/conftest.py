import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _libraries():
    """The host library and the oracle are built on demand (seconds); the HIP library must already exist."""
    import subprocess
    pkg = os.path.join(ROOT, "ray-tracing_amd")
    if not os.path.exists(os.path.join(pkg, "lib", "libv4ray_host.so")):
        subprocess.run(["make", "-s", "-C", pkg, "host"], check=True)
    if not os.path.exists(os.path.join(ROOT, "oracle", "libv4r_oracle.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    yield
