"""The C-ABI libraries load and export every entry point that include/*.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from v4ray_hip import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECL = re.compile(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b(v4rh?_\w+)\s*\(", re.M)


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = {m.group(1) for m in DECL.finditer(text)}
    return sorted(n for n in names if not n.isupper())


def test_device_library_exports_every_declared_symbol():
    names = declared("v4ray_hip.h")
    assert len(names) >= 19 and "v4r_renderer_create" in names and "v4r_render" in names
    path = os.path.join(_abi.LIB_DIR, "libv4ray_hip.so")
    assert os.path.exists(path), "build the HIP library first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = C.CDLL(path)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/v4ray_hip.h but not exported"
    assert sorted(_abi.DEVICE_SYMBOLS) == names
    lib.v4r_abi_version.restype = C.c_uint32
    assert lib.v4r_abi_version() == _abi.V4R_ABI_VERSION


def test_host_library_exports_every_declared_symbol():
    names = declared("v4ray_host.h")
    assert "v4rh_scene_flatten" in names and len(names) >= 35
    lib = C.CDLL(os.path.join(_abi.LIB_DIR, "libv4ray_host.so"))
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/v4ray_host.h but not exported"


def test_struct_sizes_match_the_header():
    assert C.sizeof(_abi.BvhNode) == 64 and C.sizeof(_abi.Prim) == 48 and C.sizeof(_abi.Xform) == 96
    assert C.sizeof(_abi.TriShading) == 64 and C.sizeof(_abi.Medium) == 16
    assert C.sizeof(_abi.Material) == 32 and C.sizeof(_abi.Texture) == 32 and C.sizeof(_abi.Image) == 16
    assert C.sizeof(_abi.Perlin) == 256 * 12 + 768 and C.sizeof(_abi.Camera) == 84
    assert C.sizeof(_abi.Stats) == 72


def test_device_entry_points_reject_null_without_a_gpu():
    lib = _abi.device_lib()
    assert lib.v4r_render(None, 0, 0, 1, 1) == -1
    assert lib.v4r_accum_read(None, None, None) == -1
    assert lib.v4r_last_error() == b"NULL renderer"
    lib.v4r_renderer_destroy(None)
    h = C.c_void_p()
    assert lib.v4r_renderer_create(None, 0, C.byref(h)) == -1 and not h.value


def test_renderer_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from v4ray_hip import Renderer, V4RayError, scenes
    s = scenes.load("c3", 16, 16)
    with pytest.raises(V4RayError, match="no HIP device|v4r error"):
        Renderer(s.renderer_param, s.camera_param, s)
