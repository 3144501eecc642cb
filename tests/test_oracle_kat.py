"""Known-answer tests that pin the CPU oracle to closed forms derived from the reference's formulas.

The reference ships no tests or vectors, so these are the pins: each case states the reference lines
(under the reference tree's src/) whose arithmetic fixes the expected value.
"""
import math

import numpy as np
import pytest

import oracle
from v4ray_hip import (Checker, Cuboid, Dielectric, DiffuseLight, ImageTexture, Lambertian, Metal, Noise,
                       PerspectiveCameraParam, RendererParam, Scene, SolidColor, Sphere, Triangle, XYRect, YZRect,
                       ZXRect, _abi)

INF = float("inf")
PARAM = RendererParam(8, 8, max_depth=5, antialias=False)
CAM = PerspectiveCameraParam((0, 0, -5), (0, 0, 0), 40.0)
GREY = Lambertian(SolidColor((0.5, 0.5, 0.5)))


def flat(scene, flags=0):
    scene.set_options(False, flags)
    return scene.flatten(PARAM, CAM)


def one(shape, material=GREY, important=False, flags=0):
    s = Scene((0, 0, 0))
    s.add(shape, material, important)
    return s, flat(s, flags)


# ---- (1),(2) sphere: hittable/sphere.rs:38-45,60-93 ---------------------------------------------------
def test_sphere_front_hit():
    s, d = one(Sphere((0, 0, 0), 1.0))
    h = oracle.hit(d, 0, (0, 0, -5), (0, 0, 1))
    assert h["mask"] and h["front_face"]
    assert h["t"] == 4.0
    assert np.allclose(h["p"], (0, 0, -1)) and np.allclose(h["normal"], (0, 0, -1))
    # sphere_uv on outward normal (0,0,-1): theta = acos(0) = pi/2, phi = atan2(1, 0) + pi = 3pi/2
    assert h["u"] == pytest.approx(0.75, abs=1e-6) and h["v"] == pytest.approx(0.5, abs=1e-6)


def test_sphere_from_inside_takes_far_root_and_flips_normal():
    s, d = one(Sphere((0, 0, 0), 1.0))
    h = oracle.hit(d, 0, (0, 0, 0), (0, 0, 1))
    assert h["mask"] and not h["front_face"]
    assert h["t"] == 1.0
    assert np.allclose(h["normal"], (0, 0, -1))  # outward (0,0,1) flipped against the ray


def test_sphere_miss_and_range():
    s, d = one(Sphere((0, 0, 0), 1.0))
    assert not oracle.hit(d, 0, (0, 2, -5), (0, 0, 1))["mask"]
    assert not oracle.hit(d, 0, (0, 0, -5), (0, 0, 1), t_min=0.001, t_max=3.9)["mask"]
    assert oracle.hit(d, 0, (0, 0, -5), (0, 0, 1), t_min=4.5, t_max=INF)["t"] == 6.0  # near root out of range


# ---- (3) rectangles: hittable/aa_rect.rs:114-144 ------------------------------------------------------
@pytest.mark.parametrize("cls,axis", [(XYRect, 2), (YZRect, 0), (ZXRect, 1)])
@pytest.mark.parametrize("positive", [True, False])
def test_rect_hits(cls, axis, positive):
    s, d = one(cls(-1, 1, -2, 2, 3.0, positive))
    ia, ib = {2: (0, 1), 0: (1, 2), 1: (2, 0)}[axis]
    direction = np.zeros(3, np.float32)
    direction[axis] = 1.0
    origin = np.zeros(3, np.float32)
    origin[axis] = -1.0
    h = oracle.hit(d, 0, origin, direction)
    assert h["mask"] and h["t"] == 4.0
    assert h["u"] == 0.5 and h["v"] == 0.5
    # outward normal is +axis when positive; the ray travels along +axis, so front_face = not positive,
    # and the stored normal always opposes the ray (hittable/mod.rs:146-154)
    assert h["front_face"] == (not positive)
    expect = np.zeros(3)
    expect[axis] = -1.0
    assert np.allclose(h["normal"], expect)
    # edge-inclusive: a == a1 and b == b0 still hit
    edge = origin.copy()
    edge[ia], edge[ib] = 1.0, -2.0
    he = oracle.hit(d, 0, edge, direction)
    assert he["mask"] and he["u"] == 1.0 and he["v"] == 0.0
    out = origin.copy()
    out[ia] = 1.0001
    assert not oracle.hit(d, 0, out, direction)["mask"]
    # from the other side
    back = origin.copy()
    back[axis] = 7.0
    hb = oracle.hit(d, 0, back, -direction)
    assert hb["mask"] and hb["front_face"] == positive and np.allclose(hb["normal"], -expect)


# ---- (4) triangle: hittable/triangle.rs:56-96 ---------------------------------------------------------
TRI = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]


def test_triangle_centroid_and_backface():
    s, d = one(Triangle(TRI))
    c = (1 / 3, 1 / 3)
    h = oracle.hit(d, 0, (c[0], c[1], -2), (0, 0, 1))
    assert h["mask"] and h["t"] == pytest.approx(2.0)
    # face normal (p2-p1)x(p3-p2) = +z (json.rs:580-586); det = e12 . (d x e13) > 0 for d = +z?  d x e13 =
    # (0,0,1)x(0,1,0) = (-1,0,0); e12 . = -1 -> det < 0 -> front_face false; the normal is NOT flipped
    assert not h["front_face"]
    assert np.allclose(h["normal"], (0, 0, 1))
    hb = oracle.hit(d, 0, (c[0], c[1], 2), (0, 0, -1))
    assert hb["mask"] and hb["front_face"] and np.allclose(hb["normal"], (0, 0, 1))


def test_triangle_edges_and_parallel():
    s, d = one(Triangle(TRI))
    assert oracle.hit(d, 0, (0.5, 0.5, -1), (0, 0, 1))["mask"]       # u + v == 1 is inside (inclusive)
    assert not oracle.hit(d, 0, (0.51, 0.5, -1), (0, 0, 1))["mask"]
    assert oracle.hit(d, 0, (0, 0, -1), (0, 0, 1))["mask"]           # vertex: u = v = 0
    assert not oracle.hit(d, 0, (-0.001, 0.2, -1), (0, 0, 1))["mask"]
    assert not oracle.hit(d, 0, (0.2, 0.2, -1), (1, 0, 0))["mask"]   # parallel: det == 0


def test_triangle_interpolates_normals_and_uvs():
    n = [(0, 0, 1), (1, 0, 0), (0, 1, 0)]
    uv = [(0, 0), (1, 0), (0, 1)]
    s, d = one(Triangle(TRI, n, uv))
    h = oracle.hit(d, 0, (0.25, 0.5, -1), (0, 0, 1))
    assert h["u"] == pytest.approx(0.25) and h["v"] == pytest.approx(0.5)
    expect = np.array([0.25, 0.5, 0.25])
    assert np.allclose(h["normal"], expect / np.linalg.norm(expect), atol=1e-6)


# ---- (5) slab test: ray.rs:167-189 --------------------------------------------------------------------
def test_slab():
    lo, hi = (-1, -1, -1), (1, 1, 1)
    assert oracle.slab(lo, hi, (0, 0, -5), (0, 0, 1))
    assert not oracle.slab(lo, hi, (2, 0, -5), (0, 0, 1))          # axis-parallel, outside in x
    assert oracle.slab(lo, hi, (0.5, 0.5, 0), (0, 0, 1))            # origin inside
    assert not oracle.slab(lo, hi, (0, 0, 5), (0, 0, 1))            # box behind
    d = np.float32([1, 1, 0]) / np.float32(math.sqrt(2))
    assert not oracle.slab(lo, hi, (-3, -1, 0), d)                  # grazes the corner (-1, 1): tmin == tmax is a miss
    assert oracle.slab(lo, hi, (-3, -1.1, 0), d)
    assert not oracle.slab(lo, hi, (0, 0, -5), (0, 0, 1), 0.001, 3.0)


# ---- (6) transform: hittable/transform.rs:64-95 -------------------------------------------------------
def test_transformed_triangle_matches_world_space_triangle():
    m = np.float32([[1000, 0, 0], [0, 1000, 0], [0, 0, 1000]])
    t = np.float32([300, 0, 200])
    tri = np.float32([(0.01, 0.02, 0.03), (0.05, 0.01, 0.02), (0.02, 0.06, 0.04)])
    s1, d1 = one(Triangle(tri).with_transform(m, t))
    s2, d2 = one(Triangle(tri * 1000 + t))
    o = np.float32([320, 25, -400])
    target = (tri.mean(axis=0) * 1000 + t).astype(np.float32)
    v = (target - o) / np.linalg.norm(target - o)
    h1, h2 = oracle.hit(d1, 0, o, v), oracle.hit(d2, 0, o, v)
    assert h1["mask"] and h2["mask"]
    assert h1["t"] == pytest.approx(h2["t"], rel=1e-5)
    assert np.allclose(h1["p"], h2["p"], rtol=1e-5, atol=1e-3)
    assert np.allclose(h1["normal"], h2["normal"], atol=1e-5)
    assert h1["front_face"] == h2["front_face"]


def test_rotated_rect_normal_and_t():
    c, sn = math.cos(math.radians(30)), math.sin(math.radians(30))
    rot_y = [[c, 0, sn], [0, 1, 0], [-sn, 0, c]]
    s, d = one(XYRect(-1, 1, -1, 1, 0.0, True).with_transform(rot_y, (0, 0, 5)))
    h = oracle.hit(d, 0, (0, 0, 0), (0, 0, 1))
    assert h["mask"] and h["t"] == pytest.approx(5.0, rel=1e-6)
    # object normal (0,0,1) hit from behind -> stored normal (0,0,-1) -> world M*(0,0,-1) = (-sin, 0, -cos)
    assert np.allclose(h["normal"], (-sn, 0, -c), atol=1e-6)


# ---- (7) reflect / refract / Schlick: material/mod.rs:47-63, dielectric.rs:23-27 ----------------------
def test_reflect_refract_schlick():
    n = (0, 1, 0)
    v45 = np.float32([1, -1, 0]) / np.float32(math.sqrt(2))
    assert np.allclose(oracle.reflect((0, -1, 0), n), (0, 1, 0))
    assert np.allclose(oracle.reflect(v45, n), np.float32([1, 1, 0]) / math.sqrt(2), atol=1e-7)
    assert np.allclose(oracle.refract((0, -1, 0), n, 1 / 1.5), (0, -1, 0))
    r = oracle.refract(v45, n, 1 / 1.5)
    sin_t = math.sin(math.radians(45)) / 1.5
    assert np.allclose(r, (sin_t, -math.sqrt(1 - sin_t ** 2), 0), atol=1e-6)
    r0 = ((1 - 1.5) / (1 + 1.5)) ** 2
    assert oracle.reflectance(1.0, 1.5) == pytest.approx(r0, rel=1e-6)
    assert oracle.reflectance(0.0, 1.5) == pytest.approx(1.0, rel=1e-6)
    assert oracle.reflectance(0.5, 1.5) == pytest.approx(r0 + (1 - r0) * 0.5 ** 5, rel=1e-6)


def test_dielectric_total_internal_reflection_and_straight_through():
    s, d = one(Sphere((0, 0, 0), 1.0), Dielectric(1.5))
    # leaving the glass at a grazing angle: ratio = ir, ratio * sin > 1 -> reflect whatever the draw
    o = np.float32([0.9, 0, 0])
    v = np.float32([0.3, 0.95, 0])
    v /= np.linalg.norm(v)
    h = oracle.hit(d, 0, o, v)
    assert h["mask"] and not h["front_face"]
    for key in range(8):
        sc = oracle.shade(d, 0, o, v, h["raw"], seg_key=key)
        assert sc["scattered"] and np.allclose(sc["coef"], 1.0)
        assert np.allclose(sc["dir"], oracle.reflect(v, h["normal"]), atol=1e-6)
    # head-on from outside: reflectance = r0 = 0.04; a draw above it refracts straight through
    h2 = oracle.hit(d, 0, (0, 0, -5), (0, 0, 1))
    dirs = [oracle.shade(d, 0, (0, 0, -5), (0, 0, 1), h2["raw"], seg_key=k)["dir"] for k in range(64)]
    straight = sum(np.allclose(x, (0, 0, 1), atol=1e-6) for x in dirs)
    back = sum(np.allclose(x, (0, 0, -1), atol=1e-6) for x in dirs)
    assert straight + back == 64 and straight >= 55


# ---- (8) cosine pdf: pdf/cosine.rs:16-47 --------------------------------------------------------------
@pytest.mark.parametrize("n", [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (0.6, 0.0, 0.8)])
def test_cosine_pdf(n):
    n = np.float32(n)
    basis = oracle.onb(n)
    assert np.allclose(basis @ basis.T, np.eye(3), atol=1e-6)      # orthonormal
    assert np.allclose(basis[2], n, atol=1e-6)                      # z axis is the normal
    assert np.linalg.det(basis.astype(np.float64)) == pytest.approx(1.0, abs=1e-5)
    rng = np.random.default_rng(3)
    est = []
    for r1, r2 in rng.random((4000, 2)):
        dvec = oracle.cosine_generate(n, float(r1), float(r2))
        assert np.dot(dvec, n) >= -1e-6 and abs(np.linalg.norm(dvec) - 1) < 1e-5
        assert oracle.cosine_value(n, dvec) == pytest.approx(max(float(np.dot(dvec, n)), 0) / math.pi, abs=1e-6)
        est.append(1.0)
    # integral of value over the sphere by uniform sampling == 1
    u = rng.normal(size=(20000, 3)).astype(np.float32)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    vals = np.array([oracle.cosine_value(n, x) for x in u])
    assert vals.mean() * 4 * math.pi == pytest.approx(1.0, abs=0.03)
    assert oracle.cosine_value(n, -n) == 0.0


# ---- (9) light pdfs: aa_rect.rs:151-185, sphere.rs:100-145, triangle.rs:103-146 -----------------------
def test_rect_light_pdf_is_area_density():
    light = DiffuseLight(SolidColor((1, 1, 1)))
    s, d = one(ZXRect(100, 200, 300, 500, 554.0, False), light, important=True)
    origin = np.float32([350, 100, 140])
    for u0, u1 in [(0.1, 0.2), (0.9, 0.5), (0.5, 0.99)]:
        v = oracle.light_generate(d, 0, origin, u0, u1)
        # ZXRect draws x (the b range) first, then z (aa_rect.rs:69-75)
        point = np.float32([300 + 200 * u0, 554, 100 + 100 * u1])
        expect = (point - origin) / np.linalg.norm(point - origin)
        assert np.allclose(v, expect, atol=1e-6)
        t = np.linalg.norm(point - origin)
        pdf = oracle.light_value(d, 0, origin, v)
        area, cos = 100 * 200, abs(v[1])
        assert pdf == pytest.approx(t * t / (cos * area), rel=1e-5)
        assert pdf * area * cos / (t * t) == pytest.approx(1.0, rel=1e-5)


def test_rect_light_pdf_plane_mode_vs_bounded():
    light = DiffuseLight(SolidColor((1, 1, 1)))
    shape = ZXRect(100, 200, 300, 500, 554.0, False)
    s, d = one(shape, light, important=True)
    origin = np.float32([350, 100, 140])
    miss = np.float32([0.9, 0.43, 0.0])
    miss /= np.linalg.norm(miss)
    t = (554 - 100) / miss[1]
    # default: the un-clipped plane density a SIMD lane of the reference sees (aa_rect.rs:158-179)
    assert oracle.light_value(d, 0, origin, miss) == pytest.approx(t * t / (abs(miss[1]) * 20000), rel=1e-5)
    down = np.float32([0, -1, 0])
    assert oracle.light_value(d, 0, origin, down) == pytest.approx(454.0 ** 2 / 20000, rel=1e-5)  # negative t squared
    s2, d2 = one(shape, light, important=True, flags=_abi.V4R_FLAG_BOUNDED_LIGHT_PDF)
    assert oracle.light_value(d2, 0, origin, miss) == 0.0
    assert oracle.light_value(d2, 0, origin, down) == 0.0


def test_sphere_light_pdf_is_cone():
    light = DiffuseLight(SolidColor((1, 1, 1)))
    s, d = one(Sphere((0, 10, 0), 2.0), light, important=True)
    origin = np.float32([0, 0, 0])
    cos_max = math.sqrt(1 - 4 / 100)
    expect = 1 / (2 * math.pi * (1 - cos_max))
    rng = np.random.default_rng(5)
    for u0, u1 in rng.random((50, 2)):
        v = oracle.light_generate(d, 0, origin, float(u0), float(u1))
        assert v[1] >= cos_max - 1e-5
        assert oracle.light_value(d, 0, origin, v) == pytest.approx(expect, rel=1e-4)
    assert oracle.light_value(d, 0, origin, (1, 0, 0)) == 0.0


def test_triangle_light_generate_stays_inside_and_pdf():
    light = DiffuseLight(SolidColor((1, 1, 1)))
    tri = np.float32([(0, 5, 0), (2, 5, 0), (0, 5, 2)])
    s, d = one(Triangle(tri), light, important=True, flags=_abi.V4R_FLAG_BOUNDED_LIGHT_PDF)
    origin = np.float32([0.5, 0, 0.5])
    rng = np.random.default_rng(9)
    for u0, u1 in rng.random((50, 2)):
        v = oracle.light_generate(d, 0, origin, float(u0), float(u1))
        h = oracle.hit(d, 0, origin, v)
        assert h["mask"]
        pdf = oracle.light_value(d, 0, origin, v)
        assert pdf == pytest.approx(h["t"] ** 2 / (abs(v[1]) * 2.0), rel=1e-4)  # area = 2


def test_packet_statistics_rule_of_the_light_pdf():
    """lights_value_stat, the expectation of the reference's 8-lane packets (aa_rect.rs:158-179, renderer.rs:189-217,
    mixture.rs:27-32, group.rs:91-95).  One rectangle light, a direction that misses it from a floor point:
      * a direction that hits gets the plane density whatever the packet;
      * alone on its shape (n = 1) a miss gets 0 -- the bounded density;
      * with one mate (n = 2) it gets the plane density when that mate hits the light: probability 1/2 + h/2;
      * in a full packet (n >= 8 ... full from 256 on) with probability 1 - (1/2 - h/2)^7;
      * in a shape of 12 rays the ray sits in the partial packet of 4 with probability 4/12;
    h = the light's cosine-weighted solid angle seen from the point (light_cosine_share), itself checked against its
    definition by sampling cosine-distributed directions."""
    light = DiffuseLight(SolidColor((1, 1, 1)))
    s, d = one(ZXRect(-50, 50, -50, 50, 400.0, False), light, important=True)    # 100 x 100, 400 above the origin
    origin, normal = np.float32([30, 0, -20]), np.float32([0, 1, 0])
    h = oracle.light_cosine_share(d, 0, origin, normal)
    rng = np.random.default_rng(21)
    hits, n_dirs = 0, 40000
    for r1, r2 in rng.random((n_dirs, 2)):
        v = oracle.cosine_generate(normal, float(r1), float(r2))
        hits += oracle.hit(d, 0, origin, v)["mask"]
    assert h == pytest.approx(hits / n_dirs, rel=0.1)          # centre / area / normal estimate vs the sampled share
    assert 0.01 < h < 0.05
    miss = np.float32([0.6, 0.8, 0.0])
    toward = oracle.light_generate(d, 0, origin, 0.3, 0.6)
    plane = oracle.light_value(d, 0, origin, miss)
    assert plane > 0 and not oracle.hit(d, 0, origin, miss)["mask"]
    keys = [oracle.lib().v4o_rng_segment_key(77, 0, k, 3) for k in range(4000)]
    for n in (1, 2, 12, 300):
        assert all(oracle.lights_value_stat(d, origin, normal, toward, 0, n, k) == pytest.approx(oracle.light_value(d, 0, origin, toward))
                   for k in keys[:50])
    got = {n: np.mean([oracle.lights_value_stat(d, origin, normal, miss, 0, n, k) != 0.0 for k in keys]) for n in (1, 2, 8, 12, 300)}
    vals = {oracle.lights_value_stat(d, origin, normal, miss, 0, 300, k) for k in keys[:200]}
    assert all(v == 0.0 or abs(v - plane) <= 1e-6 * plane for v in vals) and len(vals) == 2
    q = 0.5 - 0.5 * h
    tol = 3 * 0.5 / math.sqrt(len(keys))
    assert got[1] == 0.0
    assert got[2] == pytest.approx(1 - q, abs=tol)
    assert got[8] == pytest.approx(1 - q ** 7, abs=tol) and got[300] == pytest.approx(1 - q ** 7, abs=tol)
    assert got[12] == pytest.approx((4 / 12) * (1 - q ** 3) + (8 / 12) * (1 - q ** 7), abs=tol)
    # a second light that the packet did not draw is only reached by the mates' cosine directions
    s2 = Scene((0, 0, 0))
    s2.add(ZXRect(-50, 50, -50, 50, 400.0, False), light, True)
    s2.add(ZXRect(-50, 50, -50, 50, -300.0, True), light, True)     # below the floor point: never hit by its cosine directions
    s2.add(Sphere((0, 0, 0), 0.5), Lambertian(SolidColor((0.5, 0.5, 0.5))))
    d2 = s2.flatten(RendererParam(8, 8), PerspectiveCameraParam((0, 0, -5), (0, 0, 0), 40.0))
    p0, p1 = oracle.light_value(d2, 0, origin, miss), oracle.light_value(d2, 1, origin, miss)
    assert p0 == pytest.approx(plane) and p1 > 0 and abs(p1 - p0) > 0.1 * p0
    with0 = with1 = 0
    for k in keys:                                                  # the packet drew light 1
        v = oracle.lights_value_stat(d2, origin, normal, miss, 1, 300, k)
        parts = {(a, b): 0.5 * (a * p0 + b * p1) for a in (0, 1) for b in (0, 1)}
        a, b = min(parts, key=lambda ab: abs(parts[ab] - v))
        assert abs(parts[(a, b)] - v) <= 1e-5 * (p0 + p1)
        with0 += a
        with1 += b
    assert with0 / len(keys) == pytest.approx(1 - (1 - 0.5 * h) ** 7, abs=tol)    # light 0: cosine directions of the mates only
    assert with1 / len(keys) == pytest.approx(1 - 0.5 ** 7, abs=tol)              # light 1: the mates that aimed at it


def test_transformed_light_pdf_has_no_jacobian():
    # transform.rs:103-116: value() evaluates the object-space density at the object-space origin/direction
    light = DiffuseLight(SolidColor((1, 1, 1)))
    m = np.float32([[2, 0, 0], [0, 2, 0], [0, 0, 2]])
    s, d = one(ZXRect(-1, 1, -1, 1, 1.0, False).with_transform(m, (0, 0, 0)), light, important=True)
    pdf = oracle.light_value(d, 0, (0, 0, 0), (0, 1, 0))
    assert pdf == pytest.approx(1.0 ** 2 / (1.0 * 4.0), rel=1e-6)  # object space: t = 1, area = 4 (world: t = 2, area = 16)


# ---- (10) textures: texture/checker.rs:31-38, image.rs:25-73, noise.rs:41-140 -------------------------
def test_checker_sign_pattern():
    tex = Checker(SolidColor((1, 0, 0)), SolidColor((0, 1, 0)), 1.0)  # odd = red, even = green
    s, d = one(Sphere((0, 0, 0), 1.0), Lambertian(tex))
    checker = d.contents.n_textures - 1
    h = math.pi / 2
    assert np.allclose(oracle.texture_value(d, checker, 0, 0, (h, h, h)), (0, 1, 0))       # +++ -> even
    assert np.allclose(oracle.texture_value(d, checker, 0, 0, (-h, h, h)), (1, 0, 0))      # -++ -> odd
    assert np.allclose(oracle.texture_value(d, checker, 0, 0, (-h, -h, h)), (0, 1, 0))
    assert np.allclose(oracle.texture_value(d, checker, 0, 0, (0, h, h)), (1, 0, 0))       # sin = 0 is not > 0 -> odd


def test_image_texture_corners_and_flip():
    img = np.zeros((2, 3, 3), np.uint8)
    img[0, 0] = (255, 0, 0)      # top-left
    img[0, 2] = (0, 255, 0)      # top-right
    img[1, 0] = (0, 0, 255)      # bottom-left
    img[1, 2] = (51, 102, 204)   # bottom-right
    s, d = one(Sphere((0, 0, 0), 1.0), Lambertian(ImageTexture(img)))
    tex = d.contents.n_textures - 1
    p = (0, 0, 0)
    assert np.allclose(oracle.texture_value(d, tex, 0.0, 1.0, p), (1, 0, 0))   # v = 1 -> row 0 (v is flipped)
    assert np.allclose(oracle.texture_value(d, tex, 1.0, 1.0, p), (0, 1, 0))   # u = 1 -> clamped to the last column
    assert np.allclose(oracle.texture_value(d, tex, 0.0, 0.0, p), (0, 0, 1))   # v = 0 -> 1 - v = 1 -> last row
    assert np.allclose(oracle.texture_value(d, tex, 0.99, 0.2, p), np.float32([51, 102, 204]) / np.float32(255))
    assert np.allclose(oracle.texture_value(d, tex, -3.0, 7.0, p), (1, 0, 0))  # clamped to [0,1]


def perlin_reference(tab, p):
    """Independent scalar replay of Perlin::noise / perlin_interp (noise.rs:41-140) in float64."""
    vec = np.array([[tab.vec[i][k] for k in range(3)] for i in range(256)], dtype=np.float64)
    px, py, pz = np.array(tab.perm_x[:]), np.array(tab.perm_y[:]), np.array(tab.perm_z[:])
    fl = np.floor(p)
    u, v, w = p - fl
    i, j, k = (int(x) for x in fl)
    uu, vv, ww = (x * x * (3 - 2 * x) for x in (u, v, w))
    acc = 0.0
    for di in range(2):
        for dj in range(2):
            for dk in range(2):
                c = vec[px[(i + di) % 256] ^ py[(j + dj) % 256] ^ pz[(k + dk) % 256]]
                acc += ((uu if di else 1 - uu) * (vv if dj else 1 - vv) * (ww if dk else 1 - ww)
                        * np.dot(c, [u - di, v - dj, w - dk]))
    return acc


def test_noise_texture_matches_scalar_replay():
    s, d = one(Sphere((0, 0, 0), 1.0), Lambertian(Noise(0.7, 5, seed=11)))
    tex = d.contents.n_textures - 1
    tab = d.contents.perlin[0]
    rng = np.random.default_rng(2)
    for p in rng.uniform(-300, 300, size=(40, 3)):
        p32 = np.float32(p)
        q = np.float64(p32 * np.float32(0.7))
        acc, weight = 0.0, 1.0
        for _ in range(5):
            acc += weight * perlin_reference(tab, q)
            weight *= 0.5
            q = q * 2
        got = oracle.texture_value(d, tex, 0, 0, p32)
        assert got[0] == got[1] == got[2]
        assert got[0] == pytest.approx(abs(acc), abs=2e-4)
    # lattice points: every corner weight vector has a zero where the fade weight is non-zero -> 0
    assert oracle.texture_value(d, tex, 0, 0, np.float32([10, 20, 30]) / np.float32(0.7))[0] == pytest.approx(0, abs=1e-4)


def test_perlin_tables_are_permutations_of_unit_vectors():
    s, d = one(Sphere((0, 0, 0), 1.0), Lambertian(Noise(1.0, 1, seed=4)))
    s2, d2 = one(Sphere((0, 0, 0), 1.0), Lambertian(Noise(1.0, 1, seed=4)))
    s3, d3 = one(Sphere((0, 0, 0), 1.0), Lambertian(Noise(1.0, 1, seed=5)))
    tab, tab2, tab3 = d.contents.perlin[0], d2.contents.perlin[0], d3.contents.perlin[0]
    for perm in (tab.perm_x, tab.perm_y, tab.perm_z):
        assert sorted(perm[:]) == list(range(256))
    vec = np.array([[tab.vec[i][k] for k in range(3)] for i in range(256)])
    assert np.allclose(np.linalg.norm(vec, axis=1), 1.0, atol=1e-6)
    assert bytes(tab) == bytes(tab2) and bytes(tab) != bytes(tab3)


# ---- materials: lambertian.rs:36-46, metal.rs:31-46, diffuse_light.rs:19-22 + renderer.rs:231-263 -----
def test_lambertian_without_lights_uses_attenuation_and_cosine_direction():
    s, d = one(Sphere((0, 0, 0), 1.0), Lambertian(SolidColor((0.2, 0.4, 0.6))))
    h = oracle.hit(d, 0, (0, 0, -5), (0, 0, 1))
    for key in range(16):
        sc = oracle.shade(d, 0, (0, 0, -5), (0, 0, 1), h["raw"], seg_key=key)
        assert sc["scattered"] and np.allclose(sc["coef"], (0.2, 0.4, 0.6))
        assert np.dot(sc["dir"], h["normal"]) >= 0
        r1, r2 = oracle.lib().v4o_rng_draw(key, 12), oracle.lib().v4o_rng_draw(key, 13)
        assert np.allclose(sc["dir"], oracle.cosine_generate(h["normal"], r1, r2))


def test_lambertian_with_light_weights_by_pdf_over_mixture():
    s = Scene((0, 0, 0))
    s.add(ZXRect(-50, 50, -50, 50, 100.0, False), DiffuseLight(SolidColor((5, 5, 5))), important=True)
    s.add(ZXRect(-500, 500, -500, 500, 0.0, True), Lambertian(SolidColor((0.5, 0.5, 0.5))))
    d = flat(s, _abi.V4R_FLAG_BOUNDED_LIGHT_PDF)
    o, v = np.float32([0, 50, 0]), np.float32([0, -1, 0])
    h = oracle.hit(d, 1, o, v)
    assert h["mask"] and np.allclose(h["normal"], (0, 1, 0))
    seen_light = seen_cos = 0
    for key in range(64):
        sc = oracle.shade(d, 1, o, v, h["raw"], seg_key=key)
        pdf = oracle.cosine_value(h["normal"], sc["dir"])
        mix = 0.5 * oracle.light_value(d, 0, h["p"], sc["dir"]) + 0.5 * pdf
        assert np.allclose(sc["coef"], np.float32(0.5) * np.float32(pdf) / np.float32(mix), rtol=1e-5)
        if oracle.lib().v4o_rng_draw(key, 8) < 0.5:
            seen_light += 1
            assert oracle.hit(d, 0, h["p"], sc["dir"])["mask"]   # light-sampled directions reach the light
        else:
            seen_cos += 1
    assert seen_light > 10 and seen_cos > 10


def test_metal_mirror_and_fuzz():
    s, d = one(Sphere((0, 0, 0), 1.0), Metal((0.9, 0.8, 0.7), 0.0))
    o = np.float32([0, 2, -5])
    v = np.float32([0, -0.3, 1])
    v /= np.linalg.norm(v)
    h = oracle.hit(d, 0, o, v)
    sc = oracle.shade(d, 0, o, v, h["raw"], seg_key=3)
    assert sc["scattered"] and np.allclose(sc["coef"], (0.9, 0.8, 0.7))
    assert np.allclose(sc["dir"], oracle.reflect(v, h["normal"]), atol=1e-6)
    s2, d2 = one(Sphere((0, 0, 0), 1.0), Metal((0.9, 0.8, 0.7), 0.5))
    dirs = np.array([oracle.shade(d2, 0, o, v, h["raw"], seg_key=k)["dir"] for k in range(64)])
    assert np.allclose(np.linalg.norm(dirs, axis=1), 1, atol=1e-5)
    spread = np.linalg.norm(dirs - oracle.reflect(v, h["normal"]), axis=1)
    assert spread.max() <= 0.5 + 1e-3 and spread.std() > 0.05    # within the fuzz ball around the mirror direction


def test_diffuse_light_emits_on_front_face_only():
    s, d = one(ZXRect(-1, 1, -1, 1, 5.0, False), DiffuseLight(SolidColor((4, 5, 6))), important=True)
    up = oracle.hit(d, 0, (0, 0, 0), (0, 1, 0))      # outward normal is -y: hit from below = front
    sc = oracle.shade(d, 0, (0, 0, 0), (0, 1, 0), up["raw"])
    assert not sc["scattered"] and np.allclose(sc["emitted"], (4, 5, 6))
    dn = oracle.hit(d, 0, (0, 9, 0), (0, -1, 0))
    sc = oracle.shade(d, 0, (0, 9, 0), (0, -1, 0), dn["raw"])
    assert not sc["scattered"] and np.allclose(sc["emitted"], 0)


# ---- (11) traversal: candidate set contains the brute-force closest hit -------------------------------
def test_traversal_modes_agree_with_brute_force():
    rng = np.random.default_rng(1)
    s = Scene((0, 0, 0))
    for c in rng.uniform(-10, 10, size=(60, 3)):
        s.add(Sphere(tuple(c), float(rng.uniform(0.3, 1.5))), GREY)
    for c in rng.uniform(-10, 10, size=(20, 3)):
        tri = c + rng.uniform(-2, 2, size=(3, 3))
        s.add(Triangle(tri), GREY)
    s.add(Cuboid((-3, -3, -3), (2, 1, 4)), GREY)
    d = flat(s)
    n = 3000
    o = rng.uniform(-12, 12, size=(n, 3)).astype(np.float32)
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    prim0, t0 = oracle.trace_rays(d, o, v, mode=0)
    prim1, t1 = oracle.trace_rays(d, o, v, mode=1)
    assert np.array_equal(prim0, prim1) and np.array_equal(t0, t1)
    nprim = d.contents.n_prims
    for i in range(0, n, 7):
        best, best_t = 0xFFFFFFFF, INF
        for p in range(nprim):
            h = oracle.hit(d, p, o[i], v[i])
            if h["mask"] and h["t"] < best_t:
                best, best_t = p, h["t"]
        assert best == prim0[i] and (best == 0xFFFFFFFF or best_t == t0[i])
    assert (prim0 != 0xFFFFFFFF).mean() > 0.1


def test_equal_t_keeps_lowest_shape_index():
    s = Scene((0, 0, 0))
    s.add(XYRect(-1, 1, -1, 1, 2.0, True), GREY)
    s.add(XYRect(-2, 2, -2, 2, 2.0, False), Metal((1, 1, 1), 0.0))
    s.add(Sphere((5, 5, 5), 0.5), GREY)
    d = flat(s)
    prim, t = oracle.trace_rays(d, [(0, 0, 0)], [(0, 0, 1)])
    assert prim[0] == 0 and t[0] == 2.0          # renderer.rs:174: strict >, shapes visited in ascending order
    prim, t = oracle.trace_rays(d, [(1.5, 0, 0)], [(0, 0, 1)], mode=1)
    assert prim[0] == 1


# ---- (12) furnace + depth budget ----------------------------------------------------------------------
def test_furnace_exact():
    s = Scene((0.7, 0.9, 1.1))
    s.add(Sphere((0, 0, 0), 1.0), Lambertian(SolidColor((0.8, 0.5, 0.25))))
    d = s.flatten(RendererParam(32, 32, max_depth=5, antialias=False), PerspectiveCameraParam((0, 0, 4), (0, 0, 0), 40.0))
    img, _ = oracle.render(d, 1)
    prim, _ = oracle.trace_primary(d)
    assert (img[prim == 0] == np.float32((0.8, 0.5, 0.25)) * np.float32((0.7, 0.9, 1.1))).all()
    assert (img[prim != 0] == np.float32((0.7, 0.9, 1.1))).all()


def test_depth_budget_returns_environment():
    s = Scene((0.25, 0.25, 0.25), (2.0, 3.0, 4.0))
    s.add(Sphere((0, 0, 0), 1.0), Metal((0.5, 0.5, 0.5), 0.0))
    d = s.flatten(RendererParam(16, 16, max_depth=1, antialias=False), PerspectiveCameraParam((0, 0, 4), (0, 0, 0), 40.0))
    img, _ = oracle.render(d, 1)
    prim, _ = oracle.trace_primary(d)
    assert (img[prim == 0] == np.float32((1.0, 1.5, 2.0))).all() and (img[prim != 0] == np.float32(0.25)).all()


def test_constant_medium_free_path():
    """constant_medium.rs:40-85: hit iff -ln(U)/density <= chord length; t = entry + that distance."""
    from v4ray_hip import ConstantMedium, Isotropic
    s = Scene((0, 0, 0))
    s.add(ConstantMedium(Sphere((0, 0, 0), 1.0), 2.0), Isotropic(SolidColor((1, 1, 1))))
    d = flat(s)
    hits = 0
    for key in range(200):
        u = oracle.lib().v4o_rng_draw(key, 32)   # medium draw of primitive 0
        dist = -math.log(u) / 2.0 if u > 0 else INF
        h = oracle.hit(d, 0, (0, 0, -5), (0, 0, 1), seg_key=key)
        assert h["mask"] == (dist <= 2.0)
        if h["mask"]:
            hits += 1
            assert h["t"] == pytest.approx(4.0 + dist, rel=1e-5) and h["front_face"]
            assert np.allclose(h["normal"], (1, 0, 0))
    assert hits / 200 == pytest.approx(1 - math.exp(-4.0), abs=0.06)
    assert not oracle.hit(d, 0, (0, 3, -5), (0, 0, 1), seg_key=1)["mask"]


def test_random_streams_are_uniform_and_keyed():
    L = oracle.lib()
    keys = [L.v4o_rng_segment_key(0x5EED, p, x, s) for p in range(4) for x in range(50) for s in range(3)]
    assert len(set(keys)) == len(keys)
    u = np.array([L.v4o_rng_draw(k, dim) for k in keys[:300] for dim in range(16)])
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
    assert L.v4o_rng_segment_key(1, 2, 3, 4) == L.v4o_rng_segment_key(1, 2, 3, 4)
    assert L.v4o_rng_segment_key(1, 2, 3, 4) != L.v4o_rng_segment_key(2, 2, 3, 4)


def test_packet_statistics_rule_with_many_triangle_lights():
    """A mesh emitter: 32 triangle lights (more than the 16 draw dimensions of the packet-statistics rule, so lights l and
    l + 16 draw from different keys).  The rule (mode 3) must keep the mean of the reference's own packets (mode 2:
    aa_rect.rs:158-179 / triangle.rs:110-131 applied per 8-lane packet, one light index per packet, group.rs:91-95)."""
    s = Scene((0, 0, 0))
    white = Lambertian(SolidColor((0.7, 0.7, 0.7)))
    s.add(ZXRect(-8, 8, -8, 8, 0.0, True), white)            # floor
    s.add(XYRect(-8, 8, 0, 8, 8.0, False), white)            # back wall
    glow = DiffuseLight(SolidColor((6, 6, 6)))
    n = 0
    for i in range(4):                                       # a 4 x 4 grid of quads, two downward-facing triangles each, at y = 6
        for j in range(4):
            x0, z0 = -2.0 + i, -2.0 + j
            a, b, c, d = (x0, 6, z0), (x0 + 1, 6, z0), (x0 + 1, 6, z0 + 1), (x0, 6, z0 + 1)
            s.add(Triangle((a, b, c)), glow, important=True)
            s.add(Triangle((a, c, d)), glow, important=True)
            n += 2
    assert n == 32
    s.set_options(False, 0)
    desc = s.flatten(RendererParam(48, 48, max_depth=4), PerspectiveCameraParam((0, 3, -14), (0, 3, 0), 40.0))
    assert desc.contents.n_lights == 32
    spp = 192
    packets, _ = oracle.render(desc, spp, seed=11, mode=2)
    rule, _ = oracle.render(desc, spp, seed=12, mode=3)
    m2, m3 = packets.mean() / spp, rule.mean() / spp
    assert m2 > 0.01 and abs(m3 - m2) <= 0.03 * m2, (m2, m3)


@pytest.mark.parametrize("height", [6.0, 1.0, 0.3])
def test_packet_statistics_rule_with_a_large_nearby_light(height):
    """The rule estimates a packet mate's chance of hitting a light by cosine sampling from the light's centre, area and
    normal (light_cosine_share) -- a point-light form.  A 6 x 6 light 6, 1 and 0.3 units above the floor it lights (from far
    to nearly touching): the rule's image mean (mode 3) stays the mean of the reference's packets (mode 2)."""
    s = Scene((0, 0, 0))
    white = Lambertian(SolidColor((0.7, 0.7, 0.7)))
    s.add(ZXRect(-8, 8, -8, 8, 0.0, True), white)
    s.add(XYRect(-8, 8, 0, 8, 8.0, False), white)
    s.add(ZXRect(-3, 3, -3, 3, height, False), DiffuseLight(SolidColor((4, 4, 4))), important=True)
    s.set_options(False, 0)
    desc = s.flatten(RendererParam(48, 48, max_depth=4), PerspectiveCameraParam((0, 3, -14), (0, 3, 0), 40.0))
    spp = 256
    packets, _ = oracle.render(desc, spp, seed=21, mode=2)
    rule, _ = oracle.render(desc, spp, seed=22, mode=3)
    m2, m3 = packets.mean() / spp, rule.mean() / spp
    assert m2 > 0.01 and abs(m3 - m2) <= 0.03 * m2, (height, m2, m3)   # (measured: 0.1 %, 0.5 %, 2.1 %; two seeds of mode 2 differ by as much)
