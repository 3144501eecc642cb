"""ctypes mirror of include/v4ray_hip.h and include/v4ray_host.h, and the library loaders.

The device library (libv4ray_hip.so) is the only compute path: if it is missing, or no GPU is
present when a renderer is created, the call raises -- there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(os.path.dirname(_HERE), "lib")

V4R_ABI_VERSION = 1
V4R_LEAF_BIT = 0x80000000
V4R_NO_XFORM = 0xFFFFFFFF
V4R_FLAG_BOUNDED_LIGHT_PDF = 0x1
V4R_FLAG_KEEP_HOST_BVH = 0x2
V4R_FLAG_PLANE_LIGHT_PDF = 0x4

V4R_PRIM_SPHERE, V4R_PRIM_XY_RECT, V4R_PRIM_YZ_RECT, V4R_PRIM_ZX_RECT, V4R_PRIM_TRIANGLE, V4R_PRIM_MEDIUM = range(6)
V4R_PRIM_TYPE_MASK = 0xFF
V4R_PRIM_POSITIVE = 0x100
V4R_MAT_LAMBERTIAN, V4R_MAT_METAL, V4R_MAT_DIELECTRIC, V4R_MAT_DIFFUSE_LIGHT, V4R_MAT_ISOTROPIC = range(5)
V4R_TEX_SOLID, V4R_TEX_CHECKER, V4R_TEX_IMAGE, V4R_TEX_NOISE = range(4)


class BvhNode(C.Structure):
    _fields_ = [("lo0", C.c_float * 3), ("hi0", C.c_float * 3), ("lo1", C.c_float * 3), ("hi1", C.c_float * 3),
                ("child0", C.c_uint32), ("child1", C.c_uint32), ("pad", C.c_uint32 * 2)]


class Prim(C.Structure):
    _fields_ = [("f", C.c_float * 9), ("type", C.c_uint32), ("xform", C.c_uint32), ("aux", C.c_uint32)]


class Xform(C.Structure):
    _fields_ = [("fwd", C.c_float * 9), ("fwd_t", C.c_float * 3), ("inv", C.c_float * 9), ("inv_t", C.c_float * 3)]


class TriShading(C.Structure):
    _fields_ = [("n", (C.c_float * 3) * 3), ("uv", (C.c_float * 2) * 3), ("pad", C.c_float)]


class Medium(C.Structure):
    _fields_ = [("first", C.c_uint32), ("count", C.c_uint32), ("neg_inv_density", C.c_float), ("pad", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("tex", C.c_uint32), ("albedo", C.c_float * 3), ("fuzz", C.c_float),
                ("ir", C.c_float), ("pad", C.c_uint32)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("color", C.c_float * 3), ("s", C.c_float), ("a", C.c_uint32),
                ("b", C.c_uint32), ("pad", C.c_uint32)]


class Image(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32)]


class Perlin(C.Structure):
    _fields_ = [("vec", (C.c_float * 3) * 256), ("perm_x", C.c_uint8 * 256), ("perm_y", C.c_uint8 * 256),
                ("perm_z", C.c_uint8 * 256)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3),
                ("lens_radius", C.c_float), ("time0", C.c_float), ("time1", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("flags", C.c_uint32),
        ("n_nodes", C.c_uint32), ("bvh_depth", C.c_uint32), ("nodes", C.POINTER(BvhNode)),
        ("n_prims", C.c_uint32), ("n_xforms", C.c_uint32), ("prims", C.POINTER(Prim)),
        ("prim_material", C.POINTER(C.c_uint32)), ("xforms", C.POINTER(Xform)),
        ("n_tri_shading", C.c_uint32), ("n_media", C.c_uint32), ("tri_shading", C.POINTER(TriShading)),
        ("media", C.POINTER(Medium)),
        ("n_medium_prims", C.c_uint32), ("n_materials", C.c_uint32), ("medium_prims", C.POINTER(Prim)),
        ("materials", C.POINTER(Material)),
        ("n_textures", C.c_uint32), ("n_images", C.c_uint32), ("textures", C.POINTER(Texture)),
        ("images", C.POINTER(Image)), ("texels", C.POINTER(C.c_uint8)), ("texel_bytes", C.c_uint64),
        ("n_perlin", C.c_uint32), ("n_lights", C.c_uint32), ("perlin", C.POINTER(Perlin)),
        ("lights", C.POINTER(C.c_uint32)),
        ("background", C.c_float * 3), ("environment", C.c_float * 3), ("camera", Camera),
        ("width", C.c_uint32), ("height", C.c_uint32), ("max_depth", C.c_uint32), ("antialias", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [("segments", C.c_uint64), ("nodes", C.c_uint64), ("prims", C.c_uint64), ("paths", C.c_uint64),
                ("shaded", C.c_uint64), ("texel_fetches", C.c_uint64),
                ("wave_node_steps", C.c_uint64), ("wave_leaf_steps", C.c_uint64), ("ended_on_light", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class TreeInfo(C.Structure):  # v4r_tree_info
    _fields_ = [("width", C.c_uint32), ("n_nodes", C.c_uint32), ("n_top_nodes", C.c_uint32), ("n_prims", C.c_uint32),
                ("n_media_direct", C.c_uint32), ("media_in_tree", C.c_uint32), ("pad", C.c_uint32 * 2)]


class CameraParam(C.Structure):
    _fields_ = [("look_from", C.c_float * 3), ("look_at", C.c_float * 3), ("vfov", C.c_float), ("up", C.c_float * 3),
                ("aspect_ratio", C.c_float), ("aperture", C.c_float), ("focus_dist", C.c_float),
                ("time0", C.c_float), ("time1", C.c_float), ("has", C.c_uint32)]


CAM_UP, CAM_ASPECT, CAM_APERTURE, CAM_FOCUS, CAM_TIME0, CAM_TIME1 = 1, 2, 4, 8, 16, 32


class RendererParamC(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_depth", C.c_int32), ("antialias", C.c_int32)]


class HostBvhNode(C.Structure):
    _fields_ = [("parent_index", C.c_uint32), ("depth", C.c_uint32), ("leaf", C.c_uint32), ("shape_index", C.c_uint32),
                ("children_index", C.c_uint32 * 2), ("children_aabb", (C.c_float * 6) * 2)]


F3 = C.c_float * 3
_host = None
_dev = None


def _sig(fn, restype, *argtypes):
    fn.restype = restype
    fn.argtypes = list(argtypes)


def host_lib():
    """libv4ray_host.so (host-side scene pipeline)."""
    global _host
    if _host is not None:
        return _host
    path = os.path.join(LIB_DIR, "libv4ray_host.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C ray-tracing_amd host`")
    lib = C.CDLL(path)
    vp, cp, u32, u64, i32, f32 = C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float
    fp = C.POINTER(C.c_float)
    _sig(lib.v4rh_last_error, cp)
    _sig(lib.v4rh_scene_new, i32, fp, fp, C.POINTER(vp))
    _sig(lib.v4rh_scene_free, None, vp)
    _sig(lib.v4rh_scene_load_json_file, i32, cp, u64, C.POINTER(vp))
    _sig(lib.v4rh_scene_load_json_text, i32, cp, cp, u64, C.POINTER(vp))
    _sig(lib.v4rh_texture_solid_color, i32, vp, fp)
    _sig(lib.v4rh_texture_checker, i32, vp, i32, i32, f32)
    _sig(lib.v4rh_texture_image, i32, vp, C.POINTER(C.c_uint8), u32, u32)
    _sig(lib.v4rh_texture_image_file, i32, vp, cp)
    _sig(lib.v4rh_texture_noise, i32, vp, f32, u32, u64)
    _sig(lib.v4rh_material_lambertian, i32, vp, i32)
    _sig(lib.v4rh_material_metal, i32, vp, fp, f32)
    _sig(lib.v4rh_material_dielectric, i32, vp, f32)
    _sig(lib.v4rh_material_diffuse_light, i32, vp, i32)
    _sig(lib.v4rh_material_isotropic, i32, vp, i32)
    _sig(lib.v4rh_shape_sphere, vp, fp, f32)
    _sig(lib.v4rh_shape_rect, vp, u32, f32, f32, f32, f32, f32, i32)
    _sig(lib.v4rh_shape_triangle, vp, fp, fp, fp)
    _sig(lib.v4rh_shape_cuboid, vp, fp, fp)
    _sig(lib.v4rh_shape_mesh_file, vp, cp, i32, i32)
    _sig(lib.v4rh_shape_constant_medium, vp, vp, f32)
    _sig(lib.v4rh_shape_transform, i32, vp, fp, fp)
    _sig(lib.v4rh_shape_count, i32, vp)
    _sig(lib.v4rh_shape_free, None, vp)
    _sig(lib.v4rh_scene_add, i32, vp, vp, i32, i32)
    _sig(lib.v4rh_scene_len, i32, vp)
    _sig(lib.v4rh_scene_set_renderer, i32, vp, C.POINTER(RendererParamC))
    _sig(lib.v4rh_scene_get_renderer, i32, vp, C.POINTER(RendererParamC))
    _sig(lib.v4rh_scene_set_camera, i32, vp, C.POINTER(CameraParam))
    _sig(lib.v4rh_scene_get_camera, i32, vp, C.POINTER(CameraParam))
    _sig(lib.v4rh_scene_set_options, i32, vp, i32, u32)
    _sig(lib.v4rh_scene_flatten, i32, vp, C.POINTER(C.POINTER(SceneDesc)))
    _sig(lib.v4rh_bvh_build, i32, fp, u32, C.POINTER(HostBvhNode), C.POINTER(u32))
    _sig(lib.v4rh_scene_bounding_box, i32, vp, u32, fp)
    _sig(lib.v4rh_image_load, i32, cp, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint8), u64)
    _sig(lib.v4rh_tonemap_u8, i32, fp, u64, u32, C.POINTER(C.c_uint8))
    _sig(lib.v4rh_write_image, i32, cp, C.POINTER(C.c_uint8), u32, u32)
    _sig(lib.v4rh_checkpoint_save, i32, cp, fp, u32, u32, u32, u64)
    _sig(lib.v4rh_checkpoint_load, i32, cp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u64), fp, u64)
    _host = lib
    return lib


DEVICE_SYMBOLS = [
    "v4r_abi_version", "v4r_last_error", "v4r_renderer_create", "v4r_renderer_destroy", "v4r_render", "v4r_render_rows",
    "v4r_accum_bind", "v4r_reduce", "v4r_accum_clear", "v4r_accum_read", "v4r_accum_write", "v4r_accum_device", "v4r_tonemap_u8",
    "v4r_trace_primary", "v4r_trace_rays", "v4r_stats_enable", "v4r_stats_read", "v4r_synchronize",
    "v4r_stream", "v4r_pass_count", "v4r_profile_enable", "v4r_profile_read", "v4r_profile_read_groups", "v4r_render_noalloc",
    "v4r_tree_read",
]


def device_lib():
    """libv4ray_hip.so (gfx950 kernels + C ABI).  Raises if it has not been built."""
    global _dev
    if _dev is not None:
        return _dev
    # V4R_DEVICE_LIB selects another build of the same library (kernel tuning experiments)
    path = os.environ.get("V4R_DEVICE_LIB") or os.path.join(LIB_DIR, "libv4ray_hip.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: the HIP library must be built before rendering "
                           f"(`make -C ray-tracing_amd hip`); there is no CPU fallback")
    lib = C.CDLL(path)
    vp, cp, u32, u64, i32 = C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_int
    fp = C.POINTER(C.c_float)
    up = C.POINTER(u32)
    _sig(lib.v4r_abi_version, u32)
    _sig(lib.v4r_last_error, cp)
    _sig(lib.v4r_renderer_create, i32, C.POINTER(SceneDesc), i32, C.POINTER(vp))
    _sig(lib.v4r_renderer_destroy, None, vp)
    _sig(lib.v4r_render, i32, vp, u64, u32, u32, u32)
    _sig(lib.v4r_render_rows, i32, vp, u64, u32, u32, u32, u32, u32)
    _sig(lib.v4r_render_noalloc, i32, vp, u64, u32, u32, u32)
    _sig(lib.v4r_accum_bind, i32, vp, vp)
    _sig(lib.v4r_reduce, i32, vp, vp, i32)
    _sig(lib.v4r_accum_clear, i32, vp)
    _sig(lib.v4r_accum_read, i32, vp, fp, up)
    _sig(lib.v4r_accum_write, i32, vp, fp, u32)
    _sig(lib.v4r_accum_device, i32, vp, C.POINTER(vp), C.POINTER(u64))
    _sig(lib.v4r_tonemap_u8, i32, vp, u32, C.POINTER(C.c_uint8))
    _sig(lib.v4r_trace_primary, i32, vp, up, fp)
    _sig(lib.v4r_trace_rays, i32, vp, u32, fp, fp, u64, up, fp)
    _sig(lib.v4r_tree_read, i32, vp, C.POINTER(TreeInfo), vp, up, up)
    _sig(lib.v4r_stats_enable, i32, vp, i32)
    _sig(lib.v4r_stats_read, i32, vp, C.POINTER(Stats), i32)
    _sig(lib.v4r_synchronize, i32, vp)
    _sig(lib.v4r_stream, i32, vp, C.POINTER(vp))
    _sig(lib.v4r_pass_count, i32, vp, up)
    _sig(lib.v4r_profile_enable, i32, vp, i32)
    _sig(lib.v4r_profile_read, i32, vp, C.POINTER(C.c_double), C.POINTER(u64), i32)
    _sig(lib.v4r_profile_read_groups, i32, vp, C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(u64), i32)
    _dev = lib
    return lib


class V4RayError(RuntimeError):
    pass


def check_host(rc):
    if rc < 0:
        raise V4RayError(host_lib().v4rh_last_error().decode("utf-8", "replace"))
    return rc


def check_dev(rc):
    if rc != 0:
        raise V4RayError(f"v4r error {rc}: " + device_lib().v4r_last_error().decode("utf-8", "replace"))
    return rc
