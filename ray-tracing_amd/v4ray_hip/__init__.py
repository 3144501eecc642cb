"""Python twin of the reference's `v4ray` extension module, backed by the gfx950 path tracer.

The names and call signatures follow the reference's PyO3 surface (src/py.rs:62-86 and the stubs
under v4ray/*.pyi): `Scene`, `RendererParam`, `PerspectiveCameraParam`, `Renderer`, `AABB`, and the
submodules `shape`, `material`, `texture`.  `await Renderer.render()` returns one 1-sample-per-pixel
pass as an (H, W, 3) float32 array of linear radiance, row 0 = top (src/renderer.rs:450-476).

Shapes, materials and textures that the reference only reaches through its JSON schema (rectangles,
triangles, cuboids, meshes, constant media, diffuse lights, isotropic, image and noise textures) are
available here too.  All rendering runs in libv4ray_hip.so; nothing falls back to the CPU.
"""
import asyncio
import threading
import ctypes as C
import os
import sys
import types
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _abi
from ._abi import V4RayError, check_dev, check_host, device_lib, host_lib

Vec3 = Tuple[float, float, float]
__all__ = ["Scene", "RendererParam", "PerspectiveCameraParam", "Renderer", "AABB", "Ray", "HitRecord", "shape", "material",
           "texture", "install_as", "V4RayError"]
LANES = 8  # the reference's Python surface exchanges rays and hits in packets of f32x8 (src/py.rs:22)


def _f3(v: Sequence[float]):
    if len(v) != 3:
        raise ValueError("expected 3 components")
    return _abi.F3(float(v[0]), float(v[1]), float(v[2]))


class AABB:
    """src/bvh/aabb.rs:7-32"""

    def __init__(self, min: Vec3, max: Vec3):
        self._min = tuple(float(x) for x in min)
        self._max = tuple(float(x) for x in max)

    @property
    def min(self) -> Vec3:
        return self._min

    @property
    def max(self) -> Vec3:
        return self._max

    def __repr__(self):
        return f"AABB(min={self._min}, max={self._max})"


def _lanes(a, rows, what, dtype=np.float32):
    a = np.ascontiguousarray(a, dtype=dtype)
    want = (rows, LANES) if rows else (LANES,)
    if a.shape != want:
        raise ValueError(f"{what}: expected an array of shape {want}, got {a.shape}")
    return a


class Ray:
    """A packet of LANES rays (src/ray.rs:193-275): `origin` and `direction` are (3, LANES) float32 arrays, `time`
    (LANES,) float32 and `mask` (LANES,) bool.  The reference only ever creates these on its Rust side; here they can
    also be built from arrays (directions are normalised, as `From<&PyRay> for Ray` does, ray.rs:237-241)."""

    def __init__(self, origin, direction, time=None, mask=None):
        self._origin = _lanes(origin, 3, "origin")
        d = _lanes(direction, 3, "direction")
        with np.errstate(invalid="ignore", divide="ignore"):
            self._direction = (d / np.sqrt((d * d).sum(axis=0, dtype=np.float32))).astype(np.float32)
        self._time = _lanes(np.zeros(LANES, np.float32) if time is None else time, 0, "time")
        self._mask = _lanes(np.ones(LANES, bool) if mask is None else mask, 0, "mask", bool)

    @property
    def origin(self) -> np.ndarray:
        return self._origin.copy()

    @property
    def direction(self) -> np.ndarray:
        return self._direction.copy()

    @property
    def time(self) -> np.ndarray:
        return self._time.copy()

    @property
    def mask(self) -> np.ndarray:
        return self._mask.copy()

    def at(self, t) -> np.ndarray:
        """Points origin + t * direction, (3, LANES) (ray.rs:146-150)."""
        return (self._origin + self._direction * np.asarray(t, np.float32)).astype(np.float32)


class HitRecord:
    """The hits of a ray packet (src/hittable/mod.rs:18-31, src/hittable/py.rs:63-72): p and normal (3, LANES), t (LANES,),
    uv (2, LANES), front_face and mask (LANES,) bool; lanes with mask False hold zeros (HitRecord::default).  The reference's
    class is opaque to Python; the fields are readable here."""

    def __init__(self, p=None, normal=None, t=None, uv=None, front_face=None, mask=None):
        z = lambda rows: np.zeros((rows, LANES) if rows else (LANES,), np.float32)
        self.p = z(3) if p is None else _lanes(p, 3, "p")
        self.normal = z(3) if normal is None else _lanes(normal, 3, "normal")
        self.t = z(0) if t is None else _lanes(t, 0, "t")
        self.uv = z(2) if uv is None else _lanes(uv, 2, "uv")
        self.front_face = np.zeros(LANES, bool) if front_face is None else _lanes(front_face, 0, "front_face", bool)
        self.mask = np.zeros(LANES, bool) if mask is None else _lanes(mask, 0, "mask", bool)

    def __repr__(self):
        return f"HitRecord(t={self.t.tolist()}, mask={self.mask.tolist()})"


class PerspectiveCameraParam:
    """CameraParam, src/camera.rs:14-62"""

    def __init__(self, look_from: Vec3, look_at: Vec3, vfov: float, up: Optional[Vec3] = None,
                 aspect_ratio: Optional[float] = None, aperture: Optional[float] = None,
                 focus_dist: Optional[float] = None, time0: Optional[float] = None, time1: Optional[float] = None):
        self.look_from, self.look_at, self.vfov = tuple(look_from), tuple(look_at), float(vfov)
        self.up, self.aspect_ratio, self.aperture = up, aspect_ratio, aperture
        self.focus_dist, self.time0, self.time1 = focus_dist, time0, time1

    def _to_c(self) -> _abi.CameraParam:
        p = _abi.CameraParam()
        p.look_from, p.look_at, p.vfov = _f3(self.look_from), _f3(self.look_at), self.vfov
        has = 0
        if self.up is not None:
            p.up = _f3(self.up)
            has |= _abi.CAM_UP
        for name, bit in (("aspect_ratio", _abi.CAM_ASPECT), ("aperture", _abi.CAM_APERTURE),
                          ("focus_dist", _abi.CAM_FOCUS), ("time0", _abi.CAM_TIME0), ("time1", _abi.CAM_TIME1)):
            v = getattr(self, name)
            if v is not None:
                setattr(p, name, float(v))
                has |= bit
        p.has = has
        return p

    @classmethod
    def _from_c(cls, p: _abi.CameraParam) -> "PerspectiveCameraParam":
        def opt(bit, v):
            return v if p.has & bit else None
        return cls(tuple(p.look_from), tuple(p.look_at), p.vfov, opt(_abi.CAM_UP, tuple(p.up)),
                   opt(_abi.CAM_ASPECT, p.aspect_ratio), opt(_abi.CAM_APERTURE, p.aperture),
                   opt(_abi.CAM_FOCUS, p.focus_dist), opt(_abi.CAM_TIME0, p.time0), opt(_abi.CAM_TIME1, p.time1))


class RendererParam:
    """src/renderer.rs:42-70"""

    def __init__(self, width: int, height: int, max_depth: Optional[int] = None, antialias: Optional[bool] = None):
        if width <= 0 or height <= 0:
            raise ValueError("width and height must be positive")
        self.width, self.height, self.max_depth, self.antialias = int(width), int(height), max_depth, antialias

    def _to_c(self) -> _abi.RendererParamC:
        return _abi.RendererParamC(self.width, self.height, -1 if self.max_depth is None else int(self.max_depth),
                                   -1 if self.antialias is None else int(bool(self.antialias)))

    @classmethod
    def _from_c(cls, p: _abi.RendererParamC) -> "RendererParam":
        return cls(p.width, p.height, None if p.max_depth < 0 else p.max_depth,
                   None if p.antialias < 0 else bool(p.antialias))


class _Texture:
    def _realize(self, scene: "Scene") -> int:
        raise NotImplementedError


class _Material:
    def _realize(self, scene: "Scene") -> int:
        raise NotImplementedError


class _Shape:
    _transform = None
    _translate = None

    def _make(self):
        raise NotImplementedError

    def with_transform(self, matrix=None, translate=None):
        """TransformHittable wrapper as the `transform` / `translate` keys of src/json.rs:441-467."""
        self._transform = None if matrix is None else np.asarray(matrix, dtype=np.float32).reshape(3, 3)
        self._translate = None if translate is None else np.asarray(translate, dtype=np.float32).reshape(3)
        return self

    def _handle(self):
        lib = host_lib()
        h = self._make()
        if not h:
            raise V4RayError(lib.v4rh_last_error().decode())
        if self._transform is not None or self._translate is not None:
            fp = C.POINTER(C.c_float)
            m = None if self._transform is None else np.ascontiguousarray(self._transform).ctypes.data_as(fp)
            t = None if self._translate is None else np.ascontiguousarray(self._translate).ctypes.data_as(fp)
            try:
                check_host(lib.v4rh_shape_transform(h, m, t))
            except Exception:
                lib.v4rh_shape_free(h)
                raise
        return h


def _submodule(name: str) -> types.ModuleType:
    """A real, importable submodule (the reference adds `shape`, `material` and `texture` with add_submodule, src/py.rs:74-84)."""
    m = types.ModuleType(f"{__name__}.{name}")
    sys.modules[m.__name__] = m
    return m


def install_as(name: str = "v4ray") -> None:
    """Make `import <name>`, `import <name>.shape`, `from <name>.material import Metal` ... resolve to this package: the
    reference's front end (v4ray_frontend/*.py, main.py) imports the renderer as `v4ray`."""
    me = sys.modules[__name__]
    sys.modules[name] = me
    for sub in ("shape", "material", "texture"):
        sys.modules[f"{name}.{sub}"] = getattr(me, sub)


# ---- v4ray.texture ---------------------------------------------------------------------------------
texture = _submodule("texture")


class SolidColor(_Texture):
    """src/texture/solid_color.rs:8-39"""

    def __init__(self, color: Vec3):
        self.color = tuple(float(c) for c in color)

    def _realize(self, scene):
        return check_host(host_lib().v4rh_texture_solid_color(scene._h, _f3(self.color)))


class Checker(_Texture):
    """src/texture/checker.rs:14-66: texture1 is `odd`, texture2 is `even`."""

    def __init__(self, texture1: _Texture, texture2: _Texture, density: float):
        self.texture1, self.texture2, self.density = texture1, texture2, float(density)

    def _realize(self, scene):
        odd, even = scene._texture_id(self.texture1), scene._texture_id(self.texture2)
        return check_host(host_lib().v4rh_texture_checker(scene._h, odd, even, self.density))


class ImageTexture(_Texture):
    """src/texture/image.rs:9-17: from an (H, W, 3) uint8 array or a JPEG / PPM file."""

    def __init__(self, image):
        self.image = image

    def _realize(self, scene):
        lib = host_lib()
        if isinstance(self.image, (str, os.PathLike)):
            return check_host(lib.v4rh_texture_image_file(scene._h, os.fspath(self.image).encode()))
        arr = np.ascontiguousarray(self.image, dtype=np.uint8)
        if arr.ndim != 3 or arr.shape[2] != 3:
            raise ValueError("image must be (H, W, 3) uint8")
        return check_host(lib.v4rh_texture_image(scene._h, arr.ctypes.data_as(C.POINTER(C.c_uint8)), arr.shape[1], arr.shape[0]))


class Noise(_Texture):
    """src/texture/noise.rs:143-158; `seed` replaces the reference's thread_rng-seeded Perlin tables."""

    def __init__(self, scale: float, depth: int, seed: int = 1):
        self.scale, self.depth, self.seed = float(scale), int(depth), int(seed)

    def _realize(self, scene):
        return check_host(host_lib().v4rh_texture_noise(scene._h, self.scale, self.depth, self.seed))


texture.SolidColor, texture.Checker, texture.Image, texture.Noise = SolidColor, Checker, ImageTexture, Noise

# ---- v4ray.material --------------------------------------------------------------------------------
material = _submodule("material")


class Lambertian(_Material):
    """src/material/lambertian.rs:49-82"""

    def __init__(self, texture: _Texture):
        self.texture = texture

    def _realize(self, scene):
        return check_host(host_lib().v4rh_material_lambertian(scene._h, scene._texture_id(self.texture)))


class Metal(_Material):
    """src/material/metal.rs:13-56"""

    def __init__(self, albedo: Vec3, fuzz: float):
        self.albedo, self.fuzz = tuple(float(c) for c in albedo), float(fuzz)

    def _realize(self, scene):
        return check_host(host_lib().v4rh_material_metal(scene._h, _f3(self.albedo), self.fuzz))


class Dielectric(_Material):
    """src/material/dielectric.rs:11-70"""

    def __init__(self, ir: float):
        self.ir = float(ir)

    def _realize(self, scene):
        return check_host(host_lib().v4rh_material_dielectric(scene._h, self.ir))


class DiffuseLight(_Material):
    """src/material/diffuse_light.rs:7-23"""

    def __init__(self, emit: _Texture):
        self.emit = emit

    def _realize(self, scene):
        return check_host(host_lib().v4rh_material_diffuse_light(scene._h, scene._texture_id(self.emit)))


class Isotropic(_Material):
    """src/material/isotropic.rs:10-44"""

    def __init__(self, albedo: _Texture):
        self.albedo = albedo

    def _realize(self, scene):
        return check_host(host_lib().v4rh_material_isotropic(scene._h, scene._texture_id(self.albedo)))


material.Lambertian, material.Metal, material.Dielectric = Lambertian, Metal, Dielectric
material.DiffuseLight, material.Isotropic = DiffuseLight, Isotropic

# ---- v4ray.shape -----------------------------------------------------------------------------------
shape = _submodule("shape")


class Sphere(_Shape):
    """src/hittable/sphere.rs:25-36,147-165"""

    def __init__(self, center: Vec3, radius: float):
        self._center, self._radius = tuple(float(c) for c in center), float(radius)

    @property
    def center(self) -> Vec3:
        return self._center

    @property
    def radius(self) -> float:
        return self._radius

    def bounding_box(self, time0: float = 0.0, time1: float = 0.0) -> AABB:
        c, r = np.float32(self._center), np.float32(self._radius)
        return AABB(tuple(float(x) for x in (c - r)), tuple(float(x) for x in (c + r)))

    def hit(self, ray: Ray, t_min, t_max) -> HitRecord:
        """Sphere::hit on a packet (src/hittable/sphere.rs:60-92), in f32 like the reference: the nearer root inside
        [t_min, t_max] if there is one, else the farther; outward normal flipped against the ray; sphere_uv."""
        f = np.float32
        t_min, t_max = _lanes(t_min, 0, "t_min"), _lanes(t_max, 0, "t_max")
        o, d = ray._origin, ray._direction
        centre = f(self._center).reshape(3, 1)
        with np.errstate(all="ignore"):
            oc = o - centre
            half_b = (oc * d).sum(axis=0, dtype=f)
            c = (oc * oc).sum(axis=0, dtype=f) - f(self._radius) * f(self._radius)
            disc = half_b * half_b - c
            mask = ray._mask & ~(disc < 0)
            if not mask.any():
                return HitRecord()
            sqrt_d = np.sqrt(disc)
            root1, root2 = -half_b - sqrt_d, -half_b + sqrt_d
            mask1 = mask & (root1 >= t_min) & (root1 <= t_max)
            mask2 = mask & (root2 >= t_min) & (root2 <= t_max)
            mask = mask1 | mask2
            if not mask.any():
                return HitRecord()
            t = np.where(mask1, root1, root2).astype(f)
            p = (o + d * t).astype(f)
            outward = p - centre
            outward = (outward / np.sqrt((outward * outward).sum(axis=0, dtype=f))).astype(f)
            front = (d * outward).sum(axis=0, dtype=f) < 0
            normal = np.where(front, outward, -outward).astype(f)
            theta = np.arccos(-outward[1])
            phi = np.arctan2(-outward[2], outward[0]) + f(np.pi)
            uv = np.stack([phi / f(2 * np.pi), theta * f(1 / np.pi)]).astype(f)
        keep = lambda a: np.where(mask, a, 0).astype(a.dtype)   # lanes that miss carry nothing
        return HitRecord(keep(p), keep(normal), keep(t), keep(uv), front & mask, mask)

    def _make(self):
        return host_lib().v4rh_shape_sphere(_f3(self._center), self._radius)


class _Rect(_Shape):
    _type = None

    def __init__(self, a0, a1, b0, b1, k, positive=True):
        self.args = (float(a0), float(a1), float(b0), float(b1), float(k), bool(positive))

    def _make(self):
        a0, a1, b0, b1, k, pos = self.args
        return host_lib().v4rh_shape_rect(self._type, a0, a1, b0, b1, k, int(pos))


class XYRect(_Rect):
    """src/hittable/aa_rect.rs:190 -- (x0, x1, y0, y1, z, positive)"""
    _type = _abi.V4R_PRIM_XY_RECT


class YZRect(_Rect):
    """src/hittable/aa_rect.rs:191 -- (y0, y1, z0, z1, x, positive)"""
    _type = _abi.V4R_PRIM_YZ_RECT


class ZXRect(_Rect):
    """src/hittable/aa_rect.rs:192 -- (z0, z1, x0, x1, y, positive)"""
    _type = _abi.V4R_PRIM_ZX_RECT


class Triangle(_Shape):
    """src/hittable/triangle.rs:10-34; normals default to the face normal (src/json.rs:580-586)."""

    def __init__(self, vertices, normals=None, uvs=None):
        self.vertices = np.asarray(vertices, dtype=np.float32).reshape(3, 3)
        self.normals = None if normals is None else np.asarray(normals, dtype=np.float32).reshape(3, 3)
        self.uvs = None if uvs is None else np.asarray(uvs, dtype=np.float32).reshape(3, 2)

    def _make(self):
        fp = C.POINTER(C.c_float)
        v = np.ascontiguousarray(self.vertices)
        n = None if self.normals is None else np.ascontiguousarray(self.normals)
        u = None if self.uvs is None else np.ascontiguousarray(self.uvs)
        self._keep = (v, n, u)
        return host_lib().v4rh_shape_triangle(v.ctypes.data_as(fp), None if n is None else n.ctypes.data_as(fp),
                                              None if u is None else u.ctypes.data_as(fp))


class Cuboid(_Shape):
    """src/hittables/cuboid.rs:9-62: six rectangles"""

    def __init__(self, p0: Vec3, p1: Vec3):
        self.p0, self.p1 = tuple(p0), tuple(p1)

    def _make(self):
        return host_lib().v4rh_shape_cuboid(_f3(self.p0), _f3(self.p1))


class Mesh(_Shape):
    """src/hittables/obj.rs:11-105: one triangle per OBJ face; `subdivide` applies 1:4 midpoint subdivision first."""

    def __init__(self, file, model: int = -1, subdivide: int = 0):
        self.file, self.model, self.subdivide = os.fspath(file), int(model), int(subdivide)

    def _make(self):
        return host_lib().v4rh_shape_mesh_file(self.file.encode(), self.model, self.subdivide)


class ConstantMedium(_Shape):
    """src/hittable/constant_medium.rs:10-23"""

    def __init__(self, boundary: _Shape, density: float):
        self.boundary, self.density = boundary, float(density)

    def _make(self):
        lib = host_lib()
        inner = self.boundary._handle()
        try:
            return lib.v4rh_shape_constant_medium(inner, self.density)
        finally:
            lib.v4rh_shape_free(inner)


shape.Sphere, shape.XYRect, shape.YZRect, shape.ZXRect = Sphere, XYRect, YZRect, ZXRect
shape.Triangle, shape.Cuboid, shape.Mesh, shape.ConstantMedium = Triangle, Cuboid, Mesh, ConstantMedium


# ---- Scene -----------------------------------------------------------------------------------------
class Scene:
    """PyScene, src/scene.rs:94-119: Scene(background, environment=None), add(shape, material)."""

    def __init__(self, background: Vec3, environment: Optional[Vec3] = None, _handle=None):
        lib = host_lib()
        self._tex_ids = {}
        self._mat_ids = {}
        self._renderer_param = None
        self._camera_param = None
        if _handle is not None:
            self._h = _handle
            return
        h = C.c_void_p()
        check_host(lib.v4rh_scene_new(_f3(background), None if environment is None else _f3(environment), C.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                host_lib().v4rh_scene_free(h)
            except Exception:
                pass

    @classmethod
    def from_json(cls, path, perlin_seed: int = 1) -> "Scene":
        """serde_json::from_reader + build_scene (src/main.rs:31-34); also reads the renderer and camera blocks."""
        h = C.c_void_p()
        check_host(host_lib().v4rh_scene_load_json_file(os.fspath(path).encode(), perlin_seed, C.byref(h)))
        return cls._adopt(h)

    @classmethod
    def from_json_text(cls, text: str, base_dir: str = "", perlin_seed: int = 1) -> "Scene":
        h = C.c_void_p()
        check_host(host_lib().v4rh_scene_load_json_text(text.encode(), base_dir.encode(), perlin_seed, C.byref(h)))
        return cls._adopt(h)

    @classmethod
    def _adopt(cls, h):
        s = cls((0, 0, 0), _handle=h)
        lib = host_lib()
        rp, cp = _abi.RendererParamC(), _abi.CameraParam()
        check_host(lib.v4rh_scene_get_renderer(h, C.byref(rp)))
        check_host(lib.v4rh_scene_get_camera(h, C.byref(cp)))
        s._renderer_param = RendererParam._from_c(rp)
        s._camera_param = PerspectiveCameraParam._from_c(cp)
        return s

    @property
    def renderer_param(self) -> Optional[RendererParam]:
        return self._renderer_param

    @property
    def camera_param(self) -> Optional[PerspectiveCameraParam]:
        return self._camera_param

    def _texture_id(self, tex: _Texture) -> int:
        if not isinstance(tex, _Texture):
            raise TypeError("expected a v4ray texture")
        key = id(tex)
        if key not in self._tex_ids:
            self._tex_ids[key] = (tex._realize(self), tex)
        return self._tex_ids[key][0]

    def _material_id(self, mat: _Material) -> int:
        if not isinstance(mat, _Material):
            raise TypeError("expected a v4ray material")
        key = id(mat)
        if key not in self._mat_ids:
            self._mat_ids[key] = (mat._realize(self), mat)
        return self._mat_ids[key][0]

    def add(self, shape: _Shape, material: _Material, important: bool = False) -> None:
        """Scene::add (src/scene.rs:38-41); important=True is Scene::add_important (:52-61)."""
        if not isinstance(shape, _Shape):
            raise TypeError("expected a v4ray shape")
        lib = host_lib()
        mat = self._material_id(material)
        h = shape._handle()
        try:
            check_host(lib.v4rh_scene_add(self._h, h, mat, int(bool(important))))
        finally:
            lib.v4rh_shape_free(h)

    def __len__(self):
        return host_lib().v4rh_scene_len(self._h)

    def set_options(self, pretransform_triangles: bool = False, flags: int = 0) -> None:
        check_host(host_lib().v4rh_scene_set_options(self._h, int(pretransform_triangles), flags))

    def bounding_box(self, index: int) -> AABB:
        out = (C.c_float * 6)()
        check_host(host_lib().v4rh_scene_bounding_box(self._h, index, out))
        return AABB(tuple(out[0:3]), tuple(out[3:6]))

    def flatten(self, param: RendererParam, camera: PerspectiveCameraParam):
        """Build the BVH and return a pointer to the flat v4r_scene_desc (owned by this scene)."""
        lib = host_lib()
        rp, cp = param._to_c(), camera._to_c()
        check_host(lib.v4rh_scene_set_renderer(self._h, C.byref(rp)))
        check_host(lib.v4rh_scene_set_camera(self._h, C.byref(cp)))
        desc = C.POINTER(_abi.SceneDesc)()
        check_host(lib.v4rh_scene_flatten(self._h, C.byref(desc)))
        return desc


# ---- Renderer --------------------------------------------------------------------------------------
class Renderer:
    """PyRenderer, src/renderer.rs:430-477.

    `await renderer.render()` yields one distinct 1-spp pass as an (H, W, 3) float32 array, as the reference
    does; callers accumulate passes themselves (main.py:759-769).  `render(shared=True)` lets the calls in
    flight share one device job (for callers that only sum).  `render_passes` keeps the sum on
    the device instead, which is what the benchmark and the CLI use.
    """

    def __init__(self, param: RendererParam, camera: PerspectiveCameraParam, scene: Scene, device: int = 0,
                 seed: int = 0x5EED):
        self.param, self.camera, self.scene = param, camera, scene
        self.width, self.height = param.width, param.height
        self.seed = int(seed)
        self._next_pass = 0
        self._lock = threading.Lock()  # render() may run on several executor threads at once (main.py:1986-1990)
        self._pending, self._pending_lock, self._draining = [], threading.Lock(), False  # render() calls waiting for a batch
        self._desc = scene.flatten(param, camera)
        lib = device_lib()
        h = C.c_void_p()
        check_dev(lib.v4r_renderer_create(self._desc, int(device), C.byref(h)))
        self._h = h

    def close(self):
        # (waits for a render() batch in flight: the handle must not be destroyed under the executor thread that is using it)
        lock = getattr(self, "_lock", None)
        if lock is None:
            return
        with lock:
            h, self._h = getattr(self, "_h", None), None
            if h:
                device_lib().v4r_renderer_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def desc(self):
        return self._desc

    # -- reference-compatible single pass --
    def render_pass(self, pass_index: Optional[int] = None) -> np.ndarray:
        """One 1-spp pass (Renderer::render, src/renderer.rs:321-332) as (H, W, 3) float32."""
        lib = device_lib()
        with self._lock:  # clear + render + read is one unit: concurrent callers take turns, like the RwLock of renderer.rs:359
            if not self._h:
                raise V4RayError("the renderer is closed")
            if pass_index is None:
                pass_index = self._next_pass
                self._next_pass += 1
            check_dev(lib.v4r_accum_clear(self._h))
            check_dev(lib.v4r_render(self._h, self.seed, int(pass_index), 1, 1))
            return self.read_accumulator()[0]

    async def render(self, shared: bool = False) -> np.ndarray:
        """One pass, awaited (PyRenderer.render, src/renderer.rs:450-476): every call returns its own, distinct 1-spp pass
        as a fresh writable (H, W, 3) float32 array, as the reference does; calls in flight at the same time (main.py:1986-1990
        keeps os.cpu_count() of them going) take consecutive pass indices and are served one after the other.

        shared=True is the fast path for callers that only add the results up, which is all the reference's own callers do
        (State.with_render_result, main.py:759-769; RenderResult.add, renderer.rs:355-368): the n calls waiting at the same
        time are served by ONE device job of n consecutive passes, read back once, and each gets that job's mean pass (one
        read-only array shared by the n callers).  Their sum is the sum of n distinct passes; the individual images are not
        independent samples, so a caller that estimates variance from them, shows single passes or edits them in place
        must not ask for it."""
        loop = asyncio.get_running_loop()
        fut = loop.create_future()
        with self._pending_lock:
            self._pending.append((loop, fut, bool(shared)))
            start = not self._draining
            self._draining = True
        if start:
            loop.run_in_executor(None, self._drain)
        return await fut

    def _drain(self) -> None:
        """Executor thread: serve the waiting render() calls until none is left -- the shared ones of a round by one batch,
        the others by a pass each."""
        def deliver(fut, img, err):
            if fut.done():  # cancelled meanwhile
                return
            if err is None:
                fut.set_result(img)
            else:
                fut.set_exception(err)

        def hand(loop, fut, img, err):
            try:
                loop.call_soon_threadsafe(deliver, fut, img, err)
            except RuntimeError:  # that caller's loop has closed
                pass

        batch = []
        try:
            while True:
                with self._pending_lock:
                    batch, self._pending = self._pending, []
                    if not batch:
                        self._draining = False
                        return
                together = [(loop, fut) for loop, fut, shared in batch if shared]
                if together:
                    img, err = None, None
                    try:
                        img = self._render_mean(len(together))
                    except Exception as e:  # handed to every waiter of the batch
                        err = e
                    for loop, fut in together:
                        hand(loop, fut, img, err)
                for loop, fut, shared in batch:
                    if shared:
                        continue
                    img, err = None, None
                    try:
                        img = self.render_pass()
                    except Exception as e:
                        err = e
                    hand(loop, fut, img, err)
        except BaseException as e:  # never leave the flag standing: the next render() call must start a new drain
            with self._pending_lock:
                left, self._pending = self._pending, []
                self._draining = False
            for loop, fut, _ in batch + left:  # (calls already answered ignore it)
                hand(loop, fut, None, e if isinstance(e, Exception) else RuntimeError("render() interrupted"))
            raise

    def _render_mean(self, n: int) -> np.ndarray:
        lib = device_lib()
        with self._lock:
            if not self._h:
                raise V4RayError("the renderer is closed")
            first = self._next_pass
            self._next_pass += n
            check_dev(lib.v4r_accum_clear(self._h))
            check_dev(lib.v4r_render(self._h, self.seed, int(first), 1, int(n)))
            img = self.read_accumulator()[0]
        if n > 1:
            img *= np.float32(1.0 / n)
            img.setflags(write=False)
        return img

    # -- device-resident accumulation --
    def render_passes(self, n_passes: int, first_pass: int = 0, pass_stride: int = 1, seed: Optional[int] = None,
                      rows: Optional[Tuple[int, int]] = None) -> None:
        """Queue passes first_pass, first_pass + pass_stride, ... (n_passes of them); rows=(row0, n_rows) restricts
        them to a band of image rows (tile split of a multi-GPU job)."""
        seed = self.seed if seed is None else int(seed)
        if rows is None:
            check_dev(device_lib().v4r_render(self._h, seed, int(first_pass), int(pass_stride), int(n_passes)))
        else:
            check_dev(device_lib().v4r_render_rows(self._h, seed, int(first_pass), int(pass_stride), int(n_passes),
                                                   int(rows[0]), int(rows[1])))

    def render_passes_noalloc(self, n_passes: int, first_pass: int = 0, pass_stride: int = 1, seed: Optional[int] = None) -> None:
        """render_passes through the entry point that allocates nothing (queues in the library's static device scratch);
        raises V4RayError (V4R_E_LIMIT) when one pass of the image does not fit it."""
        seed = self.seed if seed is None else int(seed)
        check_dev(device_lib().v4r_render_noalloc(self._h, seed, int(first_pass), int(pass_stride), int(n_passes)))

    def clear(self) -> None:
        check_dev(device_lib().v4r_accum_clear(self._h))

    def synchronize(self) -> None:
        check_dev(device_lib().v4r_synchronize(self._h))

    def read_accumulator(self):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        n = C.c_uint32()
        check_dev(device_lib().v4r_accum_read(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)))
        return out, int(n.value)

    def write_accumulator(self, sums: np.ndarray, n_passes: int) -> None:
        """Restore the accumulator (H,W,3 fp32 sums) and its pass count, e.g. from a checkpoint."""
        a = np.ascontiguousarray(sums, dtype=np.float32)
        if a.shape != (self.height, self.width, 3):
            raise ValueError(f"accumulator must be ({self.height}, {self.width}, 3)")
        check_dev(device_lib().v4r_accum_write(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), n_passes))

    def save_checkpoint(self, path) -> int:
        """Write the accumulator, pass count, size and seed to `path`; returns the pass count."""
        sums, n = self.read_accumulator()
        save_checkpoint(path, sums, n, self.seed)
        return n

    def load_checkpoint(self, path) -> int:
        """Resume from a checkpoint of the same image size and seed; returns the passes it holds, so that
        render_passes(k, first_pass=returned) continues the job."""
        sums, n, seed = load_checkpoint(path)
        if sums.shape != (self.height, self.width, 3) or seed != self.seed:
            raise V4RayError(f"{path}: checkpoint of another job ({sums.shape[1]}x{sums.shape[0]}, seed {seed:#x})")
        self.write_accumulator(sums, n)
        return n

    def bind_accumulator(self, device_ptr: Optional[int]) -> None:
        check_dev(device_lib().v4r_accum_bind(self._h, C.c_void_p(device_ptr) if device_ptr else None))

    def accumulator_device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        check_dev(device_lib().v4r_accum_device(self._h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def stream_ptr(self) -> int:
        p = C.c_void_p()
        check_dev(device_lib().v4r_stream(self._h, C.byref(p)))
        return int(p.value or 0)

    def tonemap(self, n_passes: Optional[int] = None) -> np.ndarray:
        """RenderResult::get_raw (src/renderer.rs:369-406): (H, W, 3) uint8."""
        lib = device_lib()
        if n_passes is None:
            n = C.c_uint32()
            check_dev(lib.v4r_pass_count(self._h, C.byref(n)))
            n_passes = int(n.value)
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        check_dev(lib.v4r_tonemap_u8(self._h, int(n_passes), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def trace_primary(self):
        """Closest primitive id (0xFFFFFFFF = miss) and t of the deterministic primary rays, (H, W) each."""
        prim = np.empty((self.height, self.width), dtype=np.uint32)
        t = np.empty((self.height, self.width), dtype=np.float32)
        check_dev(device_lib().v4r_trace_primary(self._h, prim.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                 t.ctypes.data_as(C.POINTER(C.c_float))))
        return prim, t

    def trace_rays(self, origins, directions, seed: int = 0):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        if o.shape != d.shape:
            raise ValueError("origins and directions differ in shape")
        n = o.shape[0]
        prim = np.empty(n, dtype=np.uint32)
        t = np.empty(n, dtype=np.float32)
        fp = C.POINTER(C.c_float)
        check_dev(device_lib().v4r_trace_rays(self._h, n, o.ctypes.data_as(fp), d.ctypes.data_as(fp), seed,
                                              prim.ctypes.data_as(C.POINTER(C.c_uint32)), t.ctypes.data_as(fp)))
        return prim, t

    def enable_profile(self, mode=True) -> None:
        """0 / False: off; 1 / True: event pairs around the traversal launches; 2: exclusive per-kernel-group timing
        (every batch on one stream, pairs around traversal, shading and resolve launches)."""
        check_dev(device_lib().v4r_profile_enable(self._h, int(mode)))

    def read_profile_groups(self, reset: bool = False) -> dict:
        """Per kernel group ("trace", "shade", "resolve"): device milliseconds, kernel launches, event pairs."""
        ms, launches, pairs = (C.c_double * 3)(), (C.c_uint64 * 3)(), (C.c_uint64 * 3)()
        check_dev(device_lib().v4r_profile_read_groups(self._h, ms, launches, pairs, int(reset)))
        return {name: {"ms": float(ms[g]), "launches": int(launches[g]), "pairs": int(pairs[g])}
                for g, name in enumerate(("trace", "shade", "resolve"))}

    def read_profile(self, reset: bool = False):
        """(device milliseconds spent in the traversal kernel, number of its launches), from HIP events."""
        ms, n = C.c_double(), C.c_uint64()
        check_dev(device_lib().v4r_profile_read(self._h, C.byref(ms), C.byref(n), int(reset)))
        return float(ms.value), int(n.value)

    def read_tree(self) -> dict:
        """The tree the kernels traverse (v4r_tree_read): width, nodes as an (n_nodes, 32) uint32 array, the scene index of
        every device primitive and the constant media tested before the tree."""
        info = _abi.TreeInfo()
        check_dev(device_lib().v4r_tree_read(self._h, C.byref(info), None, None, None))
        nodes = np.zeros((info.n_nodes, 32), dtype=np.uint32)
        scene_index = np.zeros(info.n_prims, dtype=np.uint32)
        media_direct = np.zeros(info.n_media_direct, dtype=np.uint32)
        up = C.POINTER(C.c_uint32)
        check_dev(device_lib().v4r_tree_read(self._h, None, nodes.ctypes.data_as(C.c_void_p), scene_index.ctypes.data_as(up),
                                             media_direct.ctypes.data_as(up)))
        return {"width": int(info.width), "n_top_nodes": int(info.n_top_nodes), "media_in_tree": bool(info.media_in_tree),
                "nodes": nodes, "scene_index": scene_index, "media_direct": media_direct}

    def enable_stats(self, on: bool = True) -> None:
        check_dev(device_lib().v4r_stats_enable(self._h, int(on)))

    def read_stats(self, reset: bool = False) -> dict:
        s = _abi.Stats()
        check_dev(device_lib().v4r_stats_read(self._h, C.byref(s), int(reset)))
        return s.as_dict()


class RenderResult:
    """RenderResult, src/renderer.rs:335-406: host-side sum of passes + 8-bit tonemap."""

    def __init__(self, width: int, height: int):
        self.width, self.height = int(width), int(height)
        self._sum = np.zeros((self.height, self.width, 3), dtype=np.float32)
        self._count = 0

    def add(self, colors: np.ndarray) -> int:
        self._sum += np.asarray(colors, dtype=np.float32).reshape(self._sum.shape)
        self._count += 1
        return self._count

    def get_raw(self, last: int = 0):
        if self._count <= last:
            return None
        out = np.empty(self._sum.shape, dtype=np.uint8)
        s = np.ascontiguousarray(self._sum)
        check_host(host_lib().v4rh_tonemap_u8(s.ctypes.data_as(C.POINTER(C.c_float)), s.size, self._count,
                                              out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out, self._count


def load_image(path) -> np.ndarray:
    """Decode a baseline JPEG / binary PPM with the host library's decoder -> (H, W, 3) uint8."""
    lib = host_lib()
    w, h = C.c_uint32(), C.c_uint32()
    check_host(lib.v4rh_image_load(os.fspath(path).encode(), C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 3), dtype=np.uint8)
    check_host(lib.v4rh_image_load(os.fspath(path).encode(), C.byref(w), C.byref(h),
                                   out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size))
    return out


def save_checkpoint(path, sums: np.ndarray, n_passes: int, seed: int) -> None:
    a = np.ascontiguousarray(sums, dtype=np.float32)
    check_host(host_lib().v4rh_checkpoint_save(os.fspath(path).encode(), a.ctypes.data_as(C.POINTER(C.c_float)),
                                               a.shape[1], a.shape[0], n_passes, seed))


def load_checkpoint(path):
    """-> (sums (H,W,3) float32, n_passes, seed); raises on a truncated or corrupted file."""
    w, h, n, seed = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
    p = os.fspath(path).encode()
    check_host(host_lib().v4rh_checkpoint_load(p, C.byref(w), C.byref(h), C.byref(n), C.byref(seed), None, 0))
    sums = np.empty((h.value, w.value, 3), dtype=np.float32)
    check_host(host_lib().v4rh_checkpoint_load(p, None, None, None, None, sums.ctypes.data_as(C.POINTER(C.c_float)), sums.size))
    return sums, int(n.value), int(seed.value)


def write_image(path, rgb: np.ndarray) -> None:
    arr = np.ascontiguousarray(rgb, dtype=np.uint8)
    check_host(host_lib().v4rh_write_image(os.fspath(path).encode(), arr.ctypes.data_as(C.POINTER(C.c_uint8)),
                                           arr.shape[1], arr.shape[0]))
