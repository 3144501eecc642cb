"""ctypes binding of oracle/libv4r_oracle.so -- the CPU restatement of the reference's render loop.

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing_amd"))
from v4ray_hip import _abi  # noqa: E402

_LIB = None
FP = C.POINTER(C.c_float)
UP = C.POINTER(C.c_uint32)


def build():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ROOT, "oracle", "libv4r_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    D = C.POINTER(_abi.SceneDesc)
    u32, u64, i32, f32 = C.c_uint32, C.c_uint64, C.c_int, C.c_float
    L.v4o_last_error.restype = C.c_char_p
    L.v4o_render.argtypes = [D, u64, u32, u32, u32, FP, i32, u32, i32, C.POINTER(_abi.Stats)]
    L.v4o_trace_primary.argtypes = [D, UP, FP, i32]
    L.v4o_set_device_tree.argtypes = [C.POINTER(_abi.TreeInfo), C.c_void_p, UP, UP]
    L.v4o_camera_rays.argtypes = [D, u64, u32, i32, FP, FP]
    L.v4o_trace_rays.argtypes = [D, u32, FP, FP, u64, UP, FP, FP, i32]
    L.v4o_hit.argtypes = [D, u32, FP, FP, f32, f32, u32, FP]
    L.v4o_slab.argtypes = [FP, FP, FP, FP, f32, f32]
    L.v4o_texture_value.argtypes = [D, u32, f32, f32, FP, FP]
    L.v4o_shade.argtypes = [D, u32, FP, FP, FP, u32, FP]
    L.v4o_cosine_value.argtypes = [FP, FP]
    L.v4o_cosine_value.restype = f32
    L.v4o_cosine_generate.argtypes = [FP, f32, f32, FP]
    L.v4o_cosine_generate.restype = None
    L.v4o_onb.argtypes = [FP, FP]
    L.v4o_onb.restype = None
    L.v4o_light_value.argtypes = [D, u32, FP, FP, FP]
    L.v4o_light_generate.argtypes = [D, u32, FP, f32, f32, FP]
    L.v4o_lights_value_stat.argtypes = [D, FP, FP, FP, u32, u32, u32, FP]
    L.v4o_light_cosine_share.argtypes = [D, u32, FP, FP, FP]
    L.v4o_reflect.argtypes = [FP, FP, FP]
    L.v4o_reflect.restype = None
    L.v4o_refract.argtypes = [FP, FP, f32, FP]
    L.v4o_refract.restype = None
    L.v4o_reflectance.argtypes = [f32, f32]
    L.v4o_reflectance.restype = f32
    L.v4o_rng_segment_key.argtypes = [u64, u32, u32, u32]
    L.v4o_rng_segment_key.restype = u32
    L.v4o_rng_draw.argtypes = [u32, u32]
    L.v4o_rng_draw.restype = f32
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise RuntimeError("oracle: " + lib().v4o_last_error().decode())


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(FP)


def render(desc, n_passes, seed=0x5EED, first_pass=0, pass_stride=1, threads=0, pixel_stride=1, mode=0, accum=None):
    """Sum of n_passes 1-spp passes -> ((H, W, 3) float32, stats dict)."""
    d = desc.contents
    if accum is None:
        accum = np.zeros((d.height, d.width, 3), dtype=np.float32)
    st = _abi.Stats()
    _check(lib().v4o_render(desc, seed, first_pass, pass_stride, n_passes, accum.ctypes.data_as(FP), threads,
                            pixel_stride, mode, C.byref(st)))
    return accum, st.as_dict()


def set_device_tree(tree):
    """Hand the oracle the tree of a device renderer (Renderer.read_tree()) for mode 4; None clears it."""
    if tree is None:
        _check(lib().v4o_set_device_tree(None, None, None, None))
        return
    nodes = np.ascontiguousarray(tree["nodes"], dtype=np.uint32)
    index = np.ascontiguousarray(tree["scene_index"], dtype=np.uint32)
    media = np.ascontiguousarray(tree["media_direct"], dtype=np.uint32)
    info = _abi.TreeInfo(width=tree["width"], n_nodes=nodes.shape[0], n_top_nodes=tree["n_top_nodes"], n_prims=index.shape[0],
                         n_media_direct=media.shape[0], media_in_tree=int(tree["media_in_tree"]))
    _check(lib().v4o_set_device_tree(C.byref(info), nodes.ctypes.data_as(C.c_void_p), index.ctypes.data_as(UP), media.ctypes.data_as(UP)))


def trace_primary(desc, mode=0):
    d = desc.contents
    prim = np.empty((d.height, d.width), dtype=np.uint32)
    t = np.empty((d.height, d.width), dtype=np.float32)
    _check(lib().v4o_trace_primary(desc, prim.ctypes.data_as(UP), t.ctypes.data_as(FP), mode))
    return prim, t


def camera_rays(desc, seed=0x5EED, pass_index=0, deterministic=False):
    d = desc.contents
    o = np.empty((d.height * d.width, 3), dtype=np.float32)
    r = np.empty((d.height * d.width, 3), dtype=np.float32)
    _check(lib().v4o_camera_rays(desc, seed, pass_index, int(deterministic), o.ctypes.data_as(FP), r.ctypes.data_as(FP)))
    return o, r


def trace_rays(desc, origins, dirs, seed=0, mode=0, want_records=False):
    o, op = _f(np.reshape(origins, (-1, 3)))
    d, dp = _f(np.reshape(dirs, (-1, 3)))
    n = o.shape[0]
    prim = np.empty(n, dtype=np.uint32)
    t = np.empty(n, dtype=np.float32)
    rec = np.empty((n, 10), dtype=np.float32) if want_records else None
    _check(lib().v4o_trace_rays(desc, n, op, dp, seed, prim.ctypes.data_as(UP), t.ctypes.data_as(FP),
                                rec.ctypes.data_as(FP) if want_records else None, mode))
    return (prim, t, rec) if want_records else (prim, t)


def hit(desc, prim_index, origin, direction, t_min=0.001, t_max=float("inf"), seg_key=0):
    """-> dict(p, normal, t, u, v, front_face, mask)"""
    _, op = _f(origin)
    _, dp = _f(direction)
    rec = np.zeros(11, dtype=np.float32)
    _check(lib().v4o_hit(desc, prim_index, op, dp, t_min, t_max, seg_key, rec.ctypes.data_as(FP)))
    return dict(p=rec[0:3].copy(), normal=rec[3:6].copy(), t=float(rec[6]), u=float(rec[7]), v=float(rec[8]),
                front_face=bool(rec[9]), mask=bool(rec[10]), raw=rec)


def slab(lo, hi, origin, direction, t_min=0.001, t_max=float("inf")):
    return bool(lib().v4o_slab(_f(lo)[1], _f(hi)[1], _f(origin)[1], _f(direction)[1], t_min, t_max))


def texture_value(desc, tex, u, v, p):
    out = np.zeros(3, dtype=np.float32)
    _check(lib().v4o_texture_value(desc, tex, u, v, _f(p)[1], out.ctypes.data_as(FP)))
    return out


def shade(desc, prim_index, origin, direction, hit_rec, seg_key=0):
    """-> dict(scattered, coef, dir, emitted) for the 11-float record `hit(...)['raw']`."""
    out = np.zeros(10, dtype=np.float32)
    _check(lib().v4o_shade(desc, prim_index, _f(origin)[1], _f(direction)[1], _f(hit_rec)[1], seg_key,
                           out.ctypes.data_as(FP)))
    return dict(scattered=bool(out[0]), coef=out[1:4].copy(), dir=out[4:7].copy(), emitted=out[7:10].copy())


def cosine_value(normal, direction):
    return float(lib().v4o_cosine_value(_f(normal)[1], _f(direction)[1]))


def cosine_generate(normal, r1, r2):
    out = np.zeros(3, dtype=np.float32)
    lib().v4o_cosine_generate(_f(normal)[1], r1, r2, out.ctypes.data_as(FP))
    return out


def onb(normal):
    out = np.zeros(9, dtype=np.float32)
    lib().v4o_onb(_f(normal)[1], out.ctypes.data_as(FP))
    return out.reshape(3, 3)  # rows: x, y, z axes


def light_value(desc, prim_index, origin, direction):
    out = C.c_float()
    _check(lib().v4o_light_value(desc, prim_index, _f(origin)[1], _f(direction)[1], C.byref(out)))
    return float(out.value)


def lights_value_stat(desc, origin, normal, direction, li, n, seg_key):
    """The light list's pdf value under the packet-statistics rule (see v4o_lights_value_stat)."""
    out = C.c_float()
    _check(lib().v4o_lights_value_stat(desc, _f(origin)[1], _f(normal)[1], _f(direction)[1], li, n, seg_key, C.byref(out)))
    return float(out.value)


def light_cosine_share(desc, prim_index, p, n):
    out = C.c_float()
    _check(lib().v4o_light_cosine_share(desc, prim_index, _f(p)[1], _f(n)[1], C.byref(out)))
    return float(out.value)


def light_generate(desc, prim_index, origin, u0, u1):
    out = np.zeros(3, dtype=np.float32)
    _check(lib().v4o_light_generate(desc, prim_index, _f(origin)[1], u0, u1, out.ctypes.data_as(FP)))
    return out


def reflect(v, n):
    out = np.zeros(3, dtype=np.float32)
    lib().v4o_reflect(_f(v)[1], _f(n)[1], out.ctypes.data_as(FP))
    return out


def refract(v, n, ratio):
    out = np.zeros(3, dtype=np.float32)
    lib().v4o_refract(_f(v)[1], _f(n)[1], ratio, out.ctypes.data_as(FP))
    return out


def reflectance(cosine, ref_idx):
    return float(lib().v4o_reflectance(cosine, ref_idx))
