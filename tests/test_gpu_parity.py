"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same inputs."""
import numpy as np
import pytest

import oracle
from v4ray_hip import _abi
import os

from v4ray_hip import (Checker, ConstantMedium, Cuboid, Dielectric, DiffuseLight, ImageTexture, Isotropic, Lambertian, Metal, Noise,
                       PerspectiveCameraParam, Renderer, RendererParam, Scene, SolidColor, Sphere, Triangle, XYRect, YZRect, ZXRect,
                       scenes)

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF

SMALL = {"c1": (160, 160), "c2": (256, 144), "c3": (160, 160), "c4": (256, 144), "c5": (256, 144)}


def make(name, **kw):
    w, h = SMALL[name]
    scene = scenes.load(name, w, h, **kw)
    return scene, Renderer(scene.renderer_param, scene.camera_param, scene)


def check_primary(r):
    prim_d, t_d = r.trace_primary()
    prim_o, t_o = oracle.trace_primary(r.desc)
    same = prim_d == prim_o
    # north-star gate: primitive id exact and t within 1e-5 relative on >= 99.99 % of the rays
    hit = same & (prim_o != MISS)
    rel = np.zeros(prim_o.shape, dtype=np.float64)
    rel[hit] = np.abs(t_d[hit].astype(np.float64) - t_o[hit]) / np.abs(t_o[hit])
    ok = same & (rel <= 1e-5)
    assert ok.mean() >= 0.9999, f"only {ok.mean():.6f} of the primary rays agree (ids {same.mean():.6f}, max rel t {rel.max():.2e})"
    assert np.isinf(t_d[prim_d == MISS]).all()
    return same.mean(), rel.max()


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4", "c5"])
def test_primary_hits_match_oracle(name):
    scene, r = make(name)
    check_primary(r)
    r.close()


def test_primary_hits_pretransformed_bunny():
    # baking the mesh transform into the triangles is the same geometry: same ids, t to 1e-5
    scene, r = make("c4", pretransform_triangles=True)
    assert r.desc.contents.n_xforms == 0
    prim_d, t_d = r.trace_primary()
    scene2 = scenes.load("c4", *SMALL["c4"])
    desc2 = scene2.flatten(scene2.renderer_param, scene2.camera_param)
    prim_o, t_o = oracle.trace_primary(desc2)
    same = prim_d == prim_o
    hit = same & (prim_o != MISS)
    rel = np.abs(t_d[hit] - t_o[hit]) / np.abs(t_o[hit])
    assert same.mean() >= 0.9999 and np.quantile(rel, 0.9999) <= 1e-5
    r.close()


def scene_bounds(desc):
    d = desc.contents
    n = d.nodes[0]
    lo = np.minimum(np.float32(n.lo0[:]), np.float32(n.lo1[:]))
    hi = np.maximum(np.float32(n.hi0[:]), np.float32(n.hi1[:]))
    return lo, hi


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4", "c5"])
def test_random_rays_match_oracle(name):
    """Rays with random origins inside the scene's bounding box and uniformly random directions."""
    scene, r = make(name)
    rng = np.random.default_rng(7)
    n = 20000
    lo, hi = scene_bounds(r.desc)
    if name == "c2":  # the ground sphere makes the box huge: stay near the small spheres
        lo, hi = np.float32([-12, 0.01, -12]), np.float32([12, 3, 12])
    o = (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    prim_d, t_d = r.trace_rays(o, v, seed=99)   # the seed keys the free-path draws of constant media (c1)
    prim_o, t_o = oracle.trace_rays(r.desc, o, v, seed=99)
    same = prim_d == prim_o
    hit = same & (prim_o != MISS)
    rel = np.abs(t_d[hit] - t_o[hit]) / np.abs(t_o[hit])
    assert same.mean() >= 0.9995, same.mean()
    assert np.quantile(rel, 0.999) <= 1e-5
    assert hit.mean() > 0.3
    r.close()


PLANE, BOUNDED = _abi.V4R_FLAG_PLANE_LIGHT_PDF, _abi.V4R_FLAG_BOUNDED_LIGHT_PDF


@pytest.mark.parametrize("name,spp,flags", [("c1", 4, 0), ("c2", 4, 0), ("c3", 8, 0), ("c5", 4, 0), ("c4", 2, 0),
                                            ("c3", 8, PLANE), ("c4", 2, PLANE), ("c3", 8, BOUNDED), ("c4", 2, BOUNDED), ("c5", 4, BOUNDED)])
def test_render_matches_oracle_per_pixel(name, spp, flags):
    """Oracle and device draw the same random numbers, so they follow the same paths: almost every
    pixel agrees to rounding; the rest are paths where a rounding difference flipped a discrete
    decision.  The image means agree to well under the Monte-Carlo noise.  Every rule of the light pdf: the
    packet-statistics rule (no flag: per-shape ray counts of each pass and level on both sides), the plane density and
    the bounded density."""
    scene, r = make(name, flags=flags)
    r.render_passes(spp)
    img_d, n = r.read_accumulator()
    assert n == spp
    img_o, _ = oracle.render(r.desc, spp, seed=r.seed)
    assert np.isnan(img_d).sum() == np.isnan(img_o).sum() == 0
    scale = np.abs(img_o).max(axis=2) + 1e-2 * spp
    diff = np.abs(img_d - img_o).max(axis=2) / scale
    close = float((diff < 2e-3).mean())
    # what the pixels that differ weigh: their share of the image's radiance
    far = diff >= 2e-3
    share = float(np.abs(img_d - img_o).sum(axis=2)[far].sum() / np.abs(img_o).sum())
    print(f"{name}: pixels within 2e-3: {close:.5f}; the others hold {share:.2e} of the image sum; mean device {img_d.mean():.6f} oracle {img_o.mean():.6f}")
    assert close >= 0.998, close   # (measured 0.9989 - 1.0)
    assert share <= 2e-4, share   # (measured <= 3.3e-5)
    assert abs(img_d.mean() - img_o.mean()) <= 0.002 * abs(img_o.mean()) + 1e-6
    r.close()


def test_constant_medium_free_path_on_device():
    """constant_medium.rs:40-85 through the device traversal: a ray through a unit fog sphere of density 2
    is scattered iff -ln(U)/2 <= 2 (the chord), at t = entry + that distance; U is the draw the oracle
    makes for the same (seed, ray index)."""
    from v4ray_hip import ConstantMedium, Isotropic
    scene = Scene((0, 0, 0))
    scene.add(ConstantMedium(Sphere((0, 0, 0), 1.0), 2.0), Isotropic(SolidColor((1, 1, 1))))
    scene.add(Sphere((0, 0, 3), 1.0), Lambertian(SolidColor((0.5, 0.5, 0.5))))
    r = Renderer(RendererParam(16, 16, max_depth=4), PerspectiveCameraParam((0, 0, -5), (0, 0, 0), 40.0), scene)
    n = 4000
    o = np.tile(np.float32([0, 0, -5]), (n, 1))
    v = np.tile(np.float32([0, 0, 1]), (n, 1))
    prim_d, t_d = r.trace_rays(o, v, seed=5)
    prim_o, t_o = oracle.trace_rays(r.desc, o, v, seed=5)
    assert np.array_equal(prim_d, prim_o)
    fog = prim_d == 0
    assert abs(fog.mean() - (1 - np.exp(-4.0))) < 0.02       # P(scatter inside a chord of length 2 at density 2)
    assert (prim_d[~fog] == 1).all() and np.allclose(t_d[~fog], 7.0, rtol=1e-6)
    assert np.allclose(t_d[fog], t_o[fog], rtol=1e-5) and (t_d[fog] >= 4.0).all() and (t_d[fog] <= 6.0 + 1e-4).all()
    # the same scene rendered: isotropic scattering out of the fog, same paths as the oracle
    r.render_passes(8)
    img_d, _ = r.read_accumulator()
    img_o, _ = oracle.render(r.desc, 8, seed=r.seed)
    assert abs(img_d.mean() - img_o.mean()) <= 0.01 * abs(img_o.mean()) + 1e-6
    r.close()


def random_scene(seed, n_shapes):
    """Spheres, rects, triangles, cuboids and fog volumes of random size and place, half of them under a random
    rotation + non-uniform scale + translation; a few of them lights."""
    rng = np.random.default_rng(seed)
    scene = Scene((0.6, 0.7, 0.9))
    mats = [Lambertian(SolidColor((0.7, 0.6, 0.5))), Metal((0.8, 0.8, 0.8), 0.2), Dielectric(1.5),
            Lambertian(Checker(SolidColor((0.2, 0.3, 0.1)), SolidColor((0.9, 0.9, 0.9)), 3.0)),
            Lambertian(Noise(2.0, 5)), Lambertian(ImageTexture(os.path.join(scenes.ASSETS, "earthmap.jpg"))),
            Lambertian(Checker(Noise(1.0, 3), Checker(SolidColor((1, 0, 0)), SolidColor((0, 0, 1)), 7.0), 1.5))]

    def rot(rng):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        return (q * rng.uniform(0.5, 2.0, size=3)).astype(np.float32)

    for k in range(n_shapes):
        c = rng.uniform(-10, 10, size=3)
        kind = rng.integers(0, 6)
        if kind == 0:
            shape = Sphere(tuple(c), float(rng.uniform(0.2, 2.0)))
        elif kind == 1:
            a, b = np.sort(rng.uniform(-3, 3, size=2)) + c[0], np.sort(rng.uniform(-3, 3, size=2)) + c[1]
            shape = [XYRect, YZRect, ZXRect][rng.integers(0, 3)](float(a[0]), float(a[1] + 0.1), float(b[0]), float(b[1] + 0.1),
                                                                  float(c[2]), bool(rng.integers(0, 2)))
        elif kind == 2:
            v = c + rng.uniform(-2, 2, size=(3, 3))
            shape = Triangle([tuple(x) for x in v])
        elif kind == 3:
            shape = Cuboid(tuple(c), tuple(c + rng.uniform(0.3, 3, size=3)))
        elif kind == 4:
            shape = ConstantMedium(Sphere(tuple(c), float(rng.uniform(0.5, 2.5))), float(rng.uniform(0.2, 3.0)))
        else:
            shape = ConstantMedium(Cuboid(tuple(c), tuple(c + rng.uniform(0.5, 3, size=3))), float(rng.uniform(0.2, 3.0)))
        if rng.random() < 0.5:
            shape = shape.with_transform(rot(rng), rng.uniform(-3, 3, size=3))
        important = False
        if kind >= 4:
            mat = Isotropic(SolidColor((0.9, 0.9, 0.9)))
        elif kind in (0, 1, 2) and rng.random() < 0.15:
            mat = DiffuseLight(SolidColor((4, 4, 4)))
            important = bool(rng.integers(0, 2))  # half of the lights are sampled directly (mixture pdf)
        else:
            mat = mats[rng.integers(0, len(mats))]
        scene.add(shape, mat, important)
    return scene


@pytest.mark.parametrize("seed,n_shapes,tree", [(1, 3, None), (2, 40, None), (3, 400, None), (4, 2000, None),
                                                (2, 40, "8"), (3, 400, "8"), (4, 2000, "8"), (3, 400, "4")])
def test_random_scenes_match_oracle(seed, n_shapes, tree, monkeypatch):
    """Random mixed scenes (every primitive kind, transforms, media) against the oracle: random rays through the
    scene's box, and a few rendered passes.  With the tree the renderer chooses, and with a shape forced (overlapping
    media in an 8-wide tree exercise its never-cull mask)."""
    if tree:
        monkeypatch.setenv("V4R_WIDE", tree)
    scene = random_scene(seed, n_shapes)
    r = Renderer(RendererParam(96, 64, max_depth=12), PerspectiveCameraParam((0, 0, -30), (0, 0, 0), 50.0), scene)
    rng = np.random.default_rng(100 + seed)
    n = 20000
    o = rng.uniform(-14, 14, size=(n, 3)).astype(np.float32)
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    prim_d, t_d = r.trace_rays(o, v, seed=seed)
    prim_o, t_o = oracle.trace_rays(r.desc, o, v, seed=seed)
    same = prim_d == prim_o
    hit = same & (prim_o != MISS)
    rel = np.abs(t_d[hit] - t_o[hit]) / np.abs(t_o[hit])
    assert same.mean() >= 0.999, same.mean()
    assert np.quantile(rel, 0.999) <= 1e-5
    r.render_passes(4)
    img_d, _ = r.read_accumulator()
    img_o, _ = oracle.render(r.desc, 4, seed=r.seed)
    assert np.isnan(img_d).sum() == np.isnan(img_o).sum()
    assert abs(np.nanmean(img_d) - np.nanmean(img_o)) <= 0.02 * abs(np.nanmean(img_o)) + 1e-6
    close = np.abs(img_d - img_o).max(axis=2) <= 2e-3 * (np.abs(img_o).max(axis=2) + 0.04)
    print(f"random scene {seed}: {same.mean():.5f} of the rays, {close.mean():.4f} of the pixels agree")
    assert close.mean() >= 0.99, float(close.mean())
    r.close()


@pytest.mark.parametrize("seed,tree", [(10, None), (11, "8"), (12, "4"), (13, "8"), (14, None)])
def test_fuzzed_scenes_and_cameras_match_oracle(seed, tree, monkeypatch):
    """Five seeds of tools/fuzz_parity.py: an 800-shape random scene under a random camera (position, field of view, lens,
    odd image size, depth budget 1-30, antialiasing on or off), with the tree the renderer picks or one forced: closest
    hits of 30 000 random rays and a 4-pass image against the oracle's reference-order traversal."""
    if tree:
        monkeypatch.setenv("V4R_WIDE", tree)
    scene = random_scene(seed, 800)
    rng = np.random.default_rng(1000 + seed)
    eye = rng.normal(size=3)
    eye = tuple(float(x) for x in eye / np.linalg.norm(eye) * rng.uniform(5, 40))
    cam = PerspectiveCameraParam(eye, tuple(float(x) for x in rng.uniform(-3, 3, size=3)), float(rng.uniform(20, 90)),
                                 aperture=float(rng.choice([0.0, 0.3, 2.0])), focus_dist=float(rng.uniform(5, 30)))
    param = RendererParam(int(rng.integers(17, 97)), int(rng.integers(9, 71)), max_depth=int(rng.integers(1, 30)),
                          antialias=bool(rng.integers(0, 2)))
    r = Renderer(param, cam, scene)
    n = 30000
    o = rng.uniform(-14, 14, size=(n, 3)).astype(np.float32)
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    prim_d, t_d = r.trace_rays(o, v, seed=seed)
    prim_o, t_o = oracle.trace_rays(r.desc, o, v, seed=seed)
    same = prim_d == prim_o
    hit = same & (prim_o != MISS)
    rel = np.abs(t_d[hit] - t_o[hit]) / np.abs(t_o[hit])
    assert same.mean() >= 0.999 and np.quantile(rel, 0.999) <= 1e-5, (same.mean(), np.quantile(rel, 0.999))
    r.render_passes(4)
    img_d, _ = r.read_accumulator()
    img_o, _ = oracle.render(r.desc, 4, seed=r.seed)
    assert np.isnan(img_d).sum() == np.isnan(img_o).sum()
    assert abs(np.nanmean(img_d) - np.nanmean(img_o)) <= 0.02 * abs(np.nanmean(img_o)) + 1e-6
    print(f"fuzz seed {seed}, {r.read_tree()['width']}-wide tree, {param.width}x{param.height}, depth {param.max_depth}: ids {same.mean():.5f}, "
          f"t to {np.quantile(rel, 0.999):.1e}")
    r.close()


def lights_scene():
    """A room lit by one light of every samplable kind, all `important`: a sphere, a triangle, a rectangle under a
    rotation + non-uniform scale, and a plain rectangle; Lambertian, metal and glass objects between them."""
    scene = Scene((0.02, 0.02, 0.03))
    grey, red = Lambertian(SolidColor((0.7, 0.7, 0.7))), Lambertian(SolidColor((0.7, 0.2, 0.2)))
    scene.add(ZXRect(-6, 6, -6, 6, 0.0, True), grey)
    scene.add(XYRect(-6, 6, 0, 8, 6.0, False), Lambertian(Checker(SolidColor((0.1, 0.3, 0.1)), SolidColor((0.9, 0.9, 0.9)), 2.0)))
    scene.add(YZRect(0, 8, -6, 6, -6.0, True), red)
    scene.add(Sphere((0, 1, 0), 1.0), grey)
    scene.add(Sphere((2.5, 0.7, -1), 0.7), Metal((0.8, 0.8, 0.9), 0.05))
    scene.add(Sphere((-2.2, 0.8, -1.5), 0.8), Dielectric(1.5))
    scene.add(Cuboid((-4, 0, 1), (-2.5, 2.5, 2.5)), grey)
    scene.add(Sphere((3.5, 5.0, 2.0), 0.6), DiffuseLight(SolidColor((12, 10, 8))), True)
    scene.add(Triangle([(-4.0, 6.0, 0.0), (-2.0, 6.5, 1.5), (-3.5, 5.0, 2.5)]), DiffuseLight(SolidColor((10, 14, 10))), True)
    c, sn = np.cos(0.6), np.sin(0.6)
    m = np.float32([[1.4 * c, -sn, 0], [1.4 * sn, c, 0], [0, 0, 0.8]])
    scene.add(XYRect(-1, 1, -1, 1, 0.0, False).with_transform(m, (0.5, 5.5, 4.0)), DiffuseLight(SolidColor((8, 8, 14))), True)
    scene.add(ZXRect(-1, 1, -1, 1, 7.5, False), DiffuseLight(SolidColor((6, 6, 6))), True)
    return scene


@pytest.mark.parametrize("flags", [0, PLANE, BOUNDED])
def test_important_lights_of_every_kind_per_pixel(flags):
    """Samplable::{value, generate} of sphere, triangle, transformed-rectangle and rectangle lights (sphere.rs:100-144,
    triangle.rs:103-146, transform.rs:103-124, aa_rect.rs:151-185) through the mixture pdf, under each rule of the
    light pdf: per pixel against the oracle, same random streams."""
    scene = lights_scene()
    scene.set_options(False, flags)
    r = Renderer(RendererParam(160, 120, max_depth=8), PerspectiveCameraParam((0, 3, -11), (0, 2.5, 0), 45.0), scene)
    assert r.desc.contents.n_lights == 4
    check_primary(r)
    spp = 8
    r.render_passes(spp)
    img_d, _ = r.read_accumulator()
    img_o, _ = oracle.render(r.desc, spp, seed=r.seed)
    scale = np.abs(img_o).max(axis=2) + 1e-2 * spp
    close = float((np.abs(img_d - img_o).max(axis=2) / scale < 2e-3).mean())
    print(f"flags {flags}: pixels within 2e-3: {close:.4f}; mean device {img_d.mean():.6f} oracle {img_o.mean():.6f}")
    assert close >= 0.995, close
    assert abs(img_d.mean() - img_o.mean()) <= 0.01 * abs(img_o.mean())
    r.close()


def test_equal_t_keeps_the_lowest_index_on_the_device():
    """Two shapes at exactly the same distance: the reference's `t < closest` (renderer.rs:174) walks the candidates in
    ascending shape order, so the lower index wins; the device's ordered traversal must select the same one whatever
    the order it meets them in."""
    for order in ((0, 1), (1, 0)):
        scene = Scene((0.5, 0.5, 0.5))
        colours = [(0.9, 0.1, 0.1), (0.1, 0.1, 0.9)]
        for k in order:
            scene.add(XYRect(-2, 2, -2, 2, 0.0, False), Lambertian(SolidColor(colours[k])))
        scene.add(Sphere((0, 0, 3), 1.0), Lambertian(SolidColor((0.5, 0.5, 0.5))))   # a tree with more than one node
        scene.add(Sphere((3, 0, 0), 1.0), Lambertian(SolidColor((0.5, 0.5, 0.5))))
        scene.add(Sphere((3, 0, 0), 1.0), Metal((0.5, 0.5, 0.5), 0.0))               # coincident spheres too
        r = Renderer(RendererParam(64, 48, max_depth=3, antialias=False), PerspectiveCameraParam((0, 0, -8), (0, 0, 0), 60.0), scene)
        prim_d, t_d = r.trace_primary()
        prim_o, t_o = oracle.trace_primary(r.desc)
        assert np.array_equal(prim_d, prim_o)
        assert (prim_d == 0).sum() > 100 and (prim_d == 1).sum() == 0      # the rectangle added first
        assert (prim_d == 3).sum() > 10 and (prim_d == 4).sum() == 0       # the sphere added first
        rng = np.random.default_rng(7)
        o = np.float32([0, 0, -8]) + rng.uniform(-1, 1, size=(5000, 3)).astype(np.float32)
        v = rng.normal(size=(5000, 3)).astype(np.float32) * 0.3 + np.float32([0.1, 0, 1])
        v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
        pd, _ = r.trace_rays(o, v)
        po, _ = oracle.trace_rays(r.desc, o, v)
        assert np.array_equal(pd, po) and not np.isin(pd, (1, 4)).any()
        r.close()


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4", "c5"])
def test_full_size_images(name):
    """The configured scenes at their full sizes (800x800 ... 3840x2160: batches of tens of millions of paths, tiled
    camera-ray order, several pipelines): one pass agrees with the oracle per pixel, and 33 passes rendered in one
    call are bit-identical to the same passes rendered in calls of 1, 15 and 17."""
    w, h, _ = scenes.FULL_SIZE[name]
    scene = scenes.load(name, w, h)
    r = Renderer(scene.renderer_param, scene.camera_param, scene)
    one = r.render_pass(0).copy()
    ref, _ = oracle.render(r.desc, 1, seed=r.seed)
    close = float((np.abs(one - ref).max(axis=2) <= 2e-3 * (np.abs(ref).max(axis=2) + 0.01)).mean())
    print(f"{name} {w}x{h}: pixels within 2e-3 of the oracle: {close:.5f}; means {one.mean():.6f} {ref.mean():.6f}")
    assert close >= 0.995 and abs(one.mean() - ref.mean()) <= 0.005 * ref.mean()
    r.clear()
    r.render_passes(33)
    whole, n = r.read_accumulator()
    assert n == 33 and np.isfinite(whole).all()
    r.clear()
    for first, count in ((0, 1), (1, 15), (16, 17)):
        r.render_passes(count, first_pass=first)
    parts, n = r.read_accumulator()
    assert n == 33 and np.array_equal(parts, whole)
    r.close()


@pytest.mark.parametrize("name,width", [("c1", ""), ("c2", ""), ("c3", ""), ("c4", "4"), ("c4", "8"), ("c4 pretransformed", ""), ("c5", "")])
def test_primary_hits_at_full_size(name, width, monkeypatch):
    """The north-star gate at the sizes the configurations are quoted at (800x800, 1920x1080 with the 79 488-triangle
    bunny, 1024x1024, 3840x2160): the closest primitive of every deterministic camera ray is the oracle's, t within 1e-5,
    on >= 99.99 % of the rays (BVH::traverse, bvh/bvh.rs:182-239 + the closest-hit selection of renderer.rs:149-183).
    On the bunny scene, where triangles are smaller than pixels, both of the device's trees are forced in turn."""
    if width:
        monkeypatch.setenv("V4R_WIDE", width)
    w, h, _ = scenes.FULL_SIZE[name.split()[0]]
    scene = scenes.load(name.split()[0], w, h, pretransform_triangles="pretransformed" in name)
    r = Renderer(scene.renderer_param, scene.camera_param, scene)
    if width:
        assert r.read_tree()["width"] == int(width)
    if "pretransformed" in name:  # the bench's form of the scene, against the oracle on the scene as the reference holds it
        scene2 = scenes.load("c4", w, h)
        desc = scene2.flatten(scene2.renderer_param, scene2.camera_param)
        prim_d, t_d = r.trace_primary()
        prim_o, t_o = oracle.trace_primary(desc)
        same = prim_d == prim_o
        hit = same & (prim_o != MISS)
        rel = np.zeros(prim_o.shape)
        rel[hit] = np.abs(t_d[hit].astype(np.float64) - t_o[hit]) / np.abs(t_o[hit])
        ok = same & (rel <= 1e-5)
        agree, worst = ok.mean(), rel.max()
        assert agree >= 0.9999, (agree, same.mean(), worst)
    else:
        agree, worst = check_primary(r)
    print(f"{name} {w}x{h} tree {r.read_tree()['width']}-wide: ids agree on {agree:.6f} of {w * h} rays, max rel t {worst:.2e}")
    r.close()


def _degenerate_scenes():
    white = Lambertian(SolidColor((0.8, 0.8, 0.8)))
    fog = Isotropic(SolidColor((0.9, 0.9, 0.9)))
    empty = Scene((0.3, 0.5, 0.9))
    one = Scene((0.3, 0.5, 0.9))
    one.add(Sphere((0, 0, 0), 2.0), Metal((0.9, 0.8, 0.7), 0.1))
    only_fog = Scene((0.3, 0.5, 0.9))
    only_fog.add(ConstantMedium(Sphere((0, 0, 0), 3.0), 0.6), fog)
    fog_and_one = Scene((0.3, 0.5, 0.9))            # one non-medium primitive: the media stay in the tree
    fog_and_one.add(ConstantMedium(Sphere((0, 0, 0), 3.0), 0.6), fog)
    fog_and_one.add(ConstantMedium(Cuboid((-4, -4, -1), (-1, 4, 1)), 1.5), fog)
    fog_and_one.add(Sphere((1, 0, 0), 1.0), white)
    fog_and_two = Scene((0.3, 0.5, 0.9))            # the media are tested before the tree
    fog_and_two.add(Sphere((1, 0, 0), 1.0), white)
    fog_and_two.add(ConstantMedium(Sphere((0, 0, 0), 3.0), 0.6), fog)
    fog_and_two.add(XYRect(-5, 5, -5, 5, 4.0), DiffuseLight(SolidColor((3, 3, 3))), True)
    return {"empty": empty, "one sphere": one, "only fog": only_fog, "two fogs and a sphere": fog_and_one, "fog, sphere, light": fog_and_two}


@pytest.mark.parametrize("size", [(1, 1), (3, 5), (64, 48)])
def test_degenerate_scenes_match_oracle(size):
    """No primitive, one primitive (no tree), media only, media in the tree and beside it; images of one pixel and
    of sizes no tile divides."""
    cam = PerspectiveCameraParam((0, 1, -12), (0, 0, 0), 40.0)
    for what, scene in _degenerate_scenes().items():
        r = Renderer(RendererParam(size[0], size[1], max_depth=10), cam, scene)
        prim_d, t_d = r.trace_primary()
        prim_o, t_o = oracle.trace_primary(r.desc)
        assert np.array_equal(prim_d, prim_o), what
        hit = prim_o != MISS
        assert np.allclose(t_d[hit], t_o[hit], rtol=1e-5, atol=0), what
        r.render_passes(5)
        img_d, n = r.read_accumulator()
        img_o, _ = oracle.render(r.desc, 5, seed=r.seed)
        assert n == 5 and np.isfinite(img_d).all(), what
        close = np.abs(img_d - img_o).max(axis=2) <= 2e-3 * (np.abs(img_o).max(axis=2) + 0.05)
        assert close.mean() >= 0.97, (what, float(close.mean()))
        r.close()


def test_furnace_is_exact():
    """One convex Lambertian sphere, no lights, background B: every scattered ray escapes, so a hit pixel is
    exactly albedo * B and a miss pixel exactly B (renderer.rs:184,245-255,299-318)."""
    albedo, bg = (0.8, 0.5, 0.25), (0.7, 0.9, 1.1)
    scene = Scene(bg)
    scene.add(Sphere((0, 0, 0), 1.0), Lambertian(SolidColor(albedo)))
    param = RendererParam(64, 64, max_depth=5, antialias=False)
    cam = PerspectiveCameraParam((0, 0, 4), (0, 0, 0), 40.0)
    r = Renderer(param, cam, scene)
    prim, _ = r.trace_primary()
    img = r.render_pass(0)
    hit = prim == 0
    assert hit.any() and (~hit).any()
    expect_hit = np.float32(albedo) * np.float32(bg)
    assert (img[hit] == expect_hit).all()
    assert (img[~hit] == np.float32(bg)).all()
    r.close()


def test_depth_budget_returns_environment():
    """max_depth = 1: a ray that hits a scattering surface spends the budget and gets `environment`."""
    scene = Scene((0.25, 0.25, 0.25), (2.0, 3.0, 4.0))
    scene.add(Sphere((0, 0, 0), 1.0), Metal((0.5, 0.5, 0.5), 0.0))
    r = Renderer(RendererParam(32, 32, max_depth=1, antialias=False), PerspectiveCameraParam((0, 0, 4), (0, 0, 0), 40.0), scene)
    prim, _ = r.trace_primary()
    img = r.render_pass(0)
    assert (img[prim == 0] == np.float32([1.0, 1.5, 2.0])).all()
    assert (img[prim == MISS] == np.float32(0.25)).all()
    r.close()
    # max_depth = 0: ray_color returns `environment` before tracing anything (renderer.rs:128-130)
    r = Renderer(RendererParam(32, 32, max_depth=0, antialias=False), PerspectiveCameraParam((0, 0, 4), (0, 0, 0), 40.0), scene)
    r.render_passes(3)
    img, n = r.read_accumulator()
    img_o, _ = oracle.render(r.desc, 3, seed=r.seed)
    assert n == 3 and (img == np.float32([6.0, 9.0, 12.0])).all() and np.array_equal(img, img_o)
    r.close()


def test_batches_and_single_passes_are_bit_identical():
    scene, r = make("c3")
    r.render_passes(6)
    a, _ = r.read_accumulator()
    r.clear()
    for p in range(6):
        r.render_passes(1, first_pass=p)
    b, n = r.read_accumulator()
    assert n == 6 and np.array_equal(a, b)
    # a pass is a pure function of (seed, pass index)
    assert np.array_equal(r.render_pass(3), r.render_pass(3))
    assert not np.array_equal(r.render_pass(3), r.render_pass(4))
    r.close()


@pytest.mark.parametrize("size", [(63, 1), (64, 1), (65, 1), (1023, 1), (1024, 1), (1025, 1), (257, 255)])
def test_queue_lengths_around_a_group_and_a_pool(size):
    """A trace launch hands its queue out in pools of up to 1024 rays and groups of 64: queues one ray short of, equal to and one
    ray past those sizes (and, with 7 passes in one job, 7 times as long, cut at other places) give the oracle's image, and
    one job of 7 passes equals 7 jobs of one pass bit for bit."""
    scene = scenes.load("c3", *size)
    r = Renderer(scene.renderer_param, scene.camera_param, scene)
    r.render_passes(7)
    a, n = r.read_accumulator()
    assert n == 7
    r.clear()
    for p in range(7):
        r.render_passes(1, first_pass=p)
    b, _ = r.read_accumulator()
    assert np.array_equal(a, b)
    img_o, _ = oracle.render(r.desc, 7, seed=r.seed)
    far = np.abs(a - img_o).max(axis=2) >= 2e-3 * (np.abs(img_o).max(axis=2) + 0.07)
    assert far.sum() <= max(2, 0.003 * far.size), int(far.sum())   # (paths whose discrete decisions a rounding flipped)
    prim_d, t_d = r.trace_primary()
    prim_o, t_o = oracle.trace_primary(r.desc)
    assert np.array_equal(prim_d, prim_o)
    r.close()


@pytest.mark.parametrize("size,spp", [((37, 23), 21), ((8, 8), 70), ((5, 3), 3), ((64, 33), 16)])
def test_ray_order_covers_every_pixel_and_pass(size, spp):
    """The camera rays of a batch are queued tile by tile (pixels x passes), with leftover pixels and
    passes in plain order: whatever the image size and pass count, each (pixel, pass) is rendered once,
    so one batch equals the same passes rendered one at a time, bit for bit."""
    scene = scenes.load("c3", *size)
    r = Renderer(scene.renderer_param, scene.camera_param, scene)
    r.render_passes(spp)
    a, n = r.read_accumulator()
    assert n == spp
    r.clear()
    for p in range(spp):
        r.render_passes(1, first_pass=p)
    b, _ = r.read_accumulator()
    assert np.array_equal(a, b)
    img_o, _ = oracle.render(r.desc, spp, seed=r.seed)
    assert abs(a.mean() - img_o.mean()) <= 0.02 * abs(img_o.mean()) + 1e-6
    r.close()


def test_checkpoint_resume_is_bit_identical(tmp_path):
    """4 passes, checkpoint, a NEW renderer resumed from the file, 4 more passes == 8 passes in one go."""
    scene, r = make("c3")
    r.render_passes(8)
    whole, _ = r.read_accumulator()
    r.clear()
    r.render_passes(4)
    assert r.save_checkpoint(tmp_path / "c3.ckpt") == 4
    r.close()
    r2 = Renderer(scene.renderer_param, scene.camera_param, scene)
    done = r2.load_checkpoint(tmp_path / "c3.ckpt")
    assert done == 4
    r2.render_passes(4, first_pass=done)
    resumed, n = r2.read_accumulator()
    assert n == 8 and np.array_equal(resumed, whole)
    other = Renderer(scene.renderer_param, scene.camera_param, scene, seed=1234)
    with pytest.raises(Exception, match="another job"):
        other.load_checkpoint(tmp_path / "c3.ckpt")
    other.close()
    r2.close()


@pytest.mark.parametrize("name", ["c2", "c4", "c1"])
def test_second_trace_launch_changes_nothing(name, monkeypatch):
    """Rays that a nearly empty wave gives up are traced by a second launch, starting from the closest hit
    found so far: the image is bit-identical with and without it, whatever the threshold."""
    scene, r = make(name)
    r.render_passes(3)
    ref, _ = r.read_accumulator()
    r.close()
    for lanes, segs in (("0", "8"), ("32", "50"), ("63", "2")):
        monkeypatch.setenv("V4R_GIVE_UP_LANES", lanes)
        monkeypatch.setenv("V4R_GIVE_UP_SEGMENTS", segs)
        r2 = Renderer(scene.renderer_param, scene.camera_param, scene)
        r2.render_passes(3)
        img, _ = r2.read_accumulator()
        assert np.array_equal(img, ref), (lanes, segs)
        r2.close()


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4", "c5"])
def test_both_tree_shapes_give_the_same_image(name, monkeypatch):
    """The renderer keeps whichever of its two trees (4-wide with exact boxes, 8-wide with boxes on an 8-bit grid) does
    less work for the scene's camera rays.  The closest hit does not depend on the tree, so with either one forced the
    primary hits are the oracle's and the image is bit-identical; the 8-wide tree takes fewer node steps."""
    out = {}
    for width in ("4", "8"):
        monkeypatch.setenv("V4R_WIDE", width)
        scene, r = make(name)
        check_primary(r)
        r.enable_stats(True)
        r.render_passes(3)
        out[width] = (r.read_accumulator()[0], r.read_stats())
        r.close()
    assert np.array_equal(out["4"][0], out["8"][0])
    for field in ("segments", "paths", "shaded"):
        assert out["4"][1][field] == out["8"][1][field], field
    assert out["8"][1]["nodes"] != out["4"][1]["nodes"]
    if name in ("c3", "c4"):
        assert out["8"][1]["nodes"] < 0.7 * out["4"][1]["nodes"]


def test_six_wave_trace_launches_change_nothing(monkeypatch):
    """Large scenes on the 8-wide tree run the trace launches of the later segments in an instantiation held to 80
    registers (six waves per SIMD): same image and counts as the five-wave one."""
    out = {}
    for waves in ("5", "6"):
        monkeypatch.setenv("V4R_WIDE", "8")
        monkeypatch.setenv("V4R_TRACE_WAVES", waves)
        monkeypatch.setenv("V4R_GIVE_UP_LANES", "0")   # (which rays go to the second launch depends on scheduling)
        scene, r = make("c4")
        r.enable_stats(True)
        r.render_passes(3)
        out[waves] = (r.read_accumulator()[0], r.read_stats())
        r.close()
    assert np.array_equal(out["5"][0], out["6"][0])
    for field in ("segments", "paths", "shaded", "nodes", "prims"):
        assert out["5"][1][field] == out["6"][1][field], field


def test_scene_too_large_for_the_8_bit_grid_keeps_the_4_wide_tree(monkeypatch):
    """The 8-wide tree's plane distances are byte x 2^e x 1/d: beyond e = 24 (boxes wider than about 4e9) they could
    overflow for an axis-parallel ray, so such a scene has no 8-wide tree and forcing it changes nothing."""
    scene = Scene((0.5, 0.7, 1.0))
    scene.add(Sphere((0, -1e10, 0), 1e10), Lambertian(SolidColor((0.5, 0.5, 0.5))))
    rng = np.random.default_rng(3)
    for _ in range(60):
        c = rng.uniform(-8, 8, 3)
        scene.add(Sphere((float(c[0]), 0.4, float(c[2])), 0.4), Lambertian(SolidColor(tuple(float(x) for x in rng.random(3)))))
    cam = PerspectiveCameraParam((0, 2, -14), (0, 0.4, 0), 40.0)
    out = {}
    for width in ("4", "8"):
        monkeypatch.setenv("V4R_WIDE", width)
        r = Renderer(RendererParam(96, 64, max_depth=8), cam, scene)
        prim_d, t_d = r.trace_primary()
        prim_o, t_o = oracle.trace_primary(r.desc)
        assert (prim_d == prim_o).mean() >= 0.999
        r.enable_stats(True)
        r.render_passes(2)
        out[width] = (r.read_accumulator()[0], r.read_stats())
        r.close()
    assert np.array_equal(out["4"][0], out["8"][0])
    # (which rays a wave hands to the second launch depends on the order chunks were taken in: counts differ by a fraction
    # of a percent from run to run; the 8-wide tree would take a third fewer node steps)
    assert abs(out["4"][1]["nodes"] - out["8"][1]["nodes"]) <= 0.01 * out["4"][1]["nodes"]


@pytest.mark.parametrize("name,flags", [("c1", _abi.V4R_FLAG_PLANE_LIGHT_PDF), ("c2", 0), ("c5", _abi.V4R_FLAG_PLANE_LIGHT_PDF), ("c5", 0)])
def test_tail_kernel_changes_nothing(name, flags, monkeypatch):
    """From some segment on, one launch follows each remaining path to its end instead of one trace launch and
    one shading launch per class and segment: the image is bit-identical wherever that tail starts (1 = right
    after the camera rays, 0 = never), and so are the work counters.  (Under the packet-statistics rule of the light
    pdf -- scenes with lights, no flag -- the tail never starts before the levels whose rays are counted per shape.)"""
    scene, r = make(name, flags=flags)
    r.enable_stats(True)
    r.render_passes(3)
    ref, _ = r.read_accumulator()
    ref_stats = r.read_stats()
    r.close()
    monkeypatch.setenv("V4R_GIVE_UP_LANES", "0")   # a second trace launch re-enters the tree: node counts would differ
    full = None
    for start in ("0", "1", "2", "7", "49", "50", "1000"):
        monkeypatch.setenv("V4R_TAIL_SEGMENT", start)
        r2 = Renderer(scene.renderer_param, scene.camera_param, scene)
        r2.enable_stats(True)
        r2.render_passes(3)
        img, _ = r2.read_accumulator()
        stats = r2.read_stats()
        r2.close()
        assert np.array_equal(img, ref), start
        for field in ("segments", "paths", "shaded"):
            assert stats[field] == ref_stats[field], (start, field)
        full = full or stats
        for field in ("nodes", "prims", "texel_fetches"):
            assert stats[field] == full[field], (start, field)


def test_first_batches_of_fresh_pipelines_are_complete(monkeypatch):
    """The pipelines of a renderer get their queues at first use, while other pipelines are already running:
    nothing of that set-up may land in the middle of a batch.  (A null-stream memset of a new pipeline's
    counters once did; many launches per batch made it likely.)"""
    monkeypatch.setenv("V4R_GIVE_UP_LANES", "0")
    monkeypatch.setenv("V4R_TAIL_SEGMENT", "0")
    ref = ref_stats = None
    for _ in range(8):
        scene, r = make("c2")
        r.enable_stats(True)
        r.render_passes(3)
        img, _ = r.read_accumulator()
        stats = r.read_stats()
        r.close()
        if ref is None:
            ref, ref_stats = img, stats
        assert np.array_equal(img, ref)
        assert {k: stats[k] for k in ("segments", "nodes", "prims", "shaded")} == {k: ref_stats[k] for k in ("segments", "nodes", "prims", "shaded")}


def test_batching_knobs_change_nothing(monkeypatch):
    """Batch size, number of pipelines and the tiling of the camera-ray order decide where and when a path is
    computed, never what it computes: 37 passes of an odd-sized image are bit-identical under all of them."""
    scene = scenes.load("c2", 253, 141)
    def render():
        r = Renderer(scene.renderer_param, scene.camera_param, scene)
        r.render_passes(37)
        img, n = r.read_accumulator()
        r.close()
        assert n == 37
        return img
    ref = render()
    for knobs in ({"V4R_PIPES": "1"}, {"V4R_PIPES": "4", "V4R_BATCH_PATHS": str(253 * 141 * 5)}, {"V4R_BATCH_PATHS": "1"},
                  {"V4R_TILE_PASSES_LOG2": "0"}, {"V4R_TILE_PASSES_LOG2": "6", "V4R_BATCH_PATHS": str(253 * 141 * 64)},
                  {"V4R_PIPES": "2", "V4R_TILE_PASSES_LOG2": "2", "V4R_TAIL_SEGMENT": "3", "V4R_GIVE_UP_LANES": "30"}):
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        assert np.array_equal(render(), ref), knobs
        for k in knobs:
            monkeypatch.delenv(k)


def test_media_tested_before_the_tree_change_nothing():
    """Scenes with at most four constant media test them before the traversal, with a tree built over the other
    primitives; with the host's tree kept (V4R_FLAG_KEEP_HOST_BVH) the media stay leaves of the tree and the
    traversal carries the medium code.  The closest-hit rule does not depend on the order: same hits, bit-identical
    image.  With five media or more they always stay in the tree (the last case)."""
    cam = PerspectiveCameraParam((0, 0, -30), (0, 0, 0), 50.0)
    cases = [("c1", None)]
    for seed in (11, 12, 13, 14):
        cases.append((seed, 8))     # 2-4 media
    cases.append((13, 14))          # 6 media
    n_direct = 0
    for what, n_shapes in cases:
        results = []
        for flags in (0, _abi.V4R_FLAG_KEEP_HOST_BVH):
            if n_shapes is None:
                scene, r = make(what, flags=flags)
            else:
                scene = random_scene(what, n_shapes)
                scene.set_options(False, flags)
                r = Renderer(RendererParam(96, 64, max_depth=12), cam, scene)
            d = r.desc.contents
            n_media = int(sum(1 for i in range(d.n_prims) if (d.prims[i].type & 0xFF) == _abi.V4R_PRIM_MEDIUM))
            prim, t = r.trace_primary()
            r.render_passes(6)
            img, _ = r.read_accumulator()
            results.append((prim, t, img))
            r.close()
        n_direct += 1 <= n_media <= 4
        assert np.array_equal(results[0][0], results[1][0]), what
        assert np.array_equal(results[0][1], results[1][1]), what
        assert np.array_equal(results[0][2], results[1][2], equal_nan=True), what
    assert n_direct >= 5   # c1 and the 8-shape scenes are meant to take the direct path


def test_concurrent_render_calls_take_turns():
    """Renderer::render is re-entrant in the reference (main.rs:46-63): eight `render_pass()` calls on eight threads
    return eight different whole passes, the ones single calls return."""
    from concurrent.futures import ThreadPoolExecutor
    scene, r = make("c3")
    singles = [r.render_pass(k) for k in range(8)]
    r._next_pass = 0
    with ThreadPoolExecutor(8) as pool:
        got = list(pool.map(lambda _: r.render_pass(), range(8)))
    used = set()
    for img in got:
        match = [k for k in range(8) if k not in used and np.array_equal(img, singles[k])]
        assert match, "a concurrent render_pass() returned something that is no single pass"
        used.add(match[0])
    assert len(used) == 8
    r.close()


def test_awaited_render_calls_return_distinct_passes():
    """`await renderer.render()` (main.py:1986-1990 keeps os.cpu_count() in flight): every call returns a pass of its own,
    bit-identical to that pass rendered alone, as PyRenderer.render does (src/renderer.rs:450-476)."""
    import asyncio
    scene, r = make("c3")
    singles = [r.render_pass(k) for k in range(8)]

    async def many(n):
        return await asyncio.gather(*[r.render() for _ in range(n)])
    r._next_pass = 0
    got = asyncio.run(many(8))
    assert r._next_pass == 8 and len(got) == 8
    unused = list(range(8))
    for g in got:  # (the order in which the calls are served is not specified)
        match = [k for k in unused if np.array_equal(g, singles[k])]
        assert len(match) == 1 and g.flags.writeable
        unused.remove(match[0])
    r.close()


def test_awaited_render_calls_share_batches():
    """render(shared=True): calls waiting at the same time are served by one batch and get its mean pass, so their sum is
    the sum of as many distinct passes; a lone call gets exactly the single pass."""
    import asyncio
    scene, r = make("c3")
    singles = [r.render_pass(k) for k in range(12)]

    async def many(n):
        return await asyncio.gather(*[r.render(shared=True) for _ in range(n)])
    r._next_pass = 0
    got = asyncio.run(many(8))
    assert r._next_pass == 8 and len(got) == 8
    want = np.sum(singles[:8], axis=0, dtype=np.float64)
    have = np.sum(got, axis=0, dtype=np.float64)
    assert np.allclose(have, want, rtol=1e-5, atol=1e-5)
    lone = asyncio.run(many(1))[0]
    assert np.array_equal(lone, singles[8]) and lone.flags.writeable
    # three rounds of three, as a GUI loop would issue them: still every pass exactly once
    r._next_pass = 0
    total = np.zeros_like(want)
    for _ in range(4):
        total += np.sum(asyncio.run(many(3)), axis=0, dtype=np.float64)
    assert r._next_pass == 12
    assert np.allclose(total, np.sum(singles, axis=0, dtype=np.float64), rtol=1e-5, atol=1e-5)
    r.close()
    with pytest.raises(Exception):
        asyncio.run(many(1))   # a closed renderer refuses, it does not crash


def test_row_bands_add_up_to_the_full_image():
    """Tile split: bands of rows rendered separately (any heights, any order) are exactly the rows of the
    full-image render, because a pixel's random streams are keyed by its position in the whole image."""
    scene, r = make("c3", flags=_abi.V4R_FLAG_PLANE_LIGHT_PDF)   # a light pdf that is a function of the ray alone
    r.render_passes(5)
    whole, _ = r.read_accumulator()
    r.clear()
    h = r.height
    for row0, n_rows in ((h - 7, 7), (0, 1), (1, 63), (64, h - 71)):
        r.render_passes(5, rows=(row0, n_rows))
    bands, n = r.read_accumulator()
    assert np.array_equal(bands, whole) and n == 0      # partial images do not advance the pass count
    with pytest.raises(Exception, match="band"):
        r.render_passes(1, rows=(h - 1, 2))
    r.close()


def test_reduce_through_an_rccl_communicator():
    """v4r_reduce on a one-rank RCCL communicator made by the caller (C hosts do the same with N ranks): the
    call is accepted, runs on the renderer's stream and leaves the single rank's sums as they are."""
    import ctypes as C
    from v4ray_hip import _abi
    try:
        rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so.1", mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid, comm = UniqueId(), C.c_void_p()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    scene, r = make("c3")
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    r.render_passes(3)
    before, _ = r.read_accumulator()
    lib = _abi.device_lib()
    assert lib.v4r_reduce(r._h, comm, 0) == 0, lib.v4r_last_error()
    after, n = r.read_accumulator()
    assert n == 3 and np.array_equal(before, after)
    assert lib.v4r_reduce(r._h, None, 0) == -1
    rccl.ncclCommDestroy(comm)
    r.close()


def test_sample_split_over_ranks_matches_single_rank():
    """Rank r of N renders passes r, r+N, ...: the per-rank sums add up to the single-rank image
    (to fp32 reassociation) because pass p draws the same random streams wherever it runs."""
    scene, r = make("c5")
    r.render_passes(8)
    whole, _ = r.read_accumulator()
    parts = np.zeros_like(whole)
    for rank in range(2):
        r.clear()
        r.render_passes(4, first_pass=rank, pass_stride=2)
        parts += r.read_accumulator()[0]
    assert np.allclose(parts, whole, rtol=1e-5, atol=1e-6)
    r.close()


def test_tonemap_matches_host_arithmetic():
    from v4ray_hip import RenderResult
    scene, r = make("c3")
    res = RenderResult(r.width, r.height)
    for p in range(3):
        res.add(r.render_pass(p))
    host_u8, n = res.get_raw(0)
    r.clear()
    r.render_passes(3)
    dev_u8 = r.tonemap()
    assert n == 3
    assert np.abs(host_u8.astype(int) - dev_u8.astype(int)).max() <= 1
    assert (host_u8 == dev_u8).mean() > 0.999
    r.close()


@pytest.mark.parametrize("name", ["c2", "c4"])
def test_work_counters_match_oracle(name, monkeypatch):
    """Nodes fetched / primitives tested / segments per path, counted by the instrumented trace kernel,
    agree with the oracle running the same traversal order (they feed the roofline's algorithmic bytes)."""
    # same tree topology on both sides: the device re-splits the host's tree by default; and no second trace
    # launch, whose given-up rays repeat part of their traversal
    monkeypatch.setenv("V4R_GIVE_UP_LANES", "0")
    scene, r = make(name, flags=_abi.V4R_FLAG_KEEP_HOST_BVH)
    r.enable_stats(True)
    r.render_passes(2)
    dev = r.read_stats(reset=True)
    _, orc = oracle.render(r.desc, 2, seed=r.seed, mode=1)
    for key in ("segments", "nodes", "prims", "paths", "shaded"):
        assert abs(dev[key] - orc[key]) <= 0.02 * orc[key], (key, dev[key], orc[key])
    r.close()


@pytest.mark.parametrize("name,kw", [("c1", {}), ("c2", {}), ("c3", {}), ("c4", {}), ("c4", {"pretransform_triangles": True}), ("c5", {})])
def test_work_counters_of_the_timed_tree_match_oracle(name, kw, monkeypatch):
    """The same for the tree that is actually timed -- the device's own re-split tree, 4-wide or compressed 8-wide as the
    scene's default flags select -- which the oracle walks (mode 4) after reading it back through v4r_tree_read, with its
    own scalar restatement of the node steps.  Without the second trace launch the counts are the oracle's to a rounding
    (a ray either visits a node or it does not).  With it (the default) the rays a wave gives up start again at the root
    in the second launch, with the closest hit they had found: real work that the per-ray oracle does not have -- the
    device then counts more, never less (3-15 % more node visits at these sizes)."""
    for give_up, tolerance in (("0", 0.0005), (None, 0.25)):
        if give_up is None:
            monkeypatch.delenv("V4R_GIVE_UP_LANES", raising=False)
        else:
            monkeypatch.setenv("V4R_GIVE_UP_LANES", give_up)
        scene, r = make(name, **kw)
        tree = r.read_tree()
        oracle.set_device_tree(tree)
        try:
            r.enable_stats(True)
            r.render_passes(2)
            dev = r.read_stats(reset=True)
            _, orc = oracle.render(r.desc, 2, seed=r.seed, mode=4)
        finally:
            oracle.set_device_tree(None)
        print(name, kw, f"{tree['width']}-wide, give-up lanes {give_up or 'default'}:", {k: (dev[k], orc[k]) for k in ("segments", "nodes", "prims")})
        for key in ("segments", "paths", "shaded"):
            assert abs(dev[key] - orc[key]) <= 0.0005 * orc[key], (key, dev[key], orc[key])
        for key in ("nodes", "prims"):
            assert abs(dev[key] - orc[key]) <= tolerance * orc[key], (key, dev[key], orc[key])
            assert dev[key] >= orc[key] * (1 - 0.0005)
        r.close()


def test_tree_read_describes_the_traversed_tree():
    """v4r_tree_read: every primitive is in exactly one leaf of the exported tree, the scene indices are a permutation,
    and the oracle walking that tree (mode 4) finds the closest hits of its reference-order traversal."""
    scene, r = make("c1")
    tree = r.read_tree()
    n = r.desc.contents.n_prims
    assert sorted(tree["scene_index"].tolist()) == list(range(n))
    nodes, width = tree["nodes"], tree["width"]
    seen = np.zeros(n, dtype=int)
    for k in tree["media_direct"]:
        seen[k] += 1
    words = nodes[:, 4:12] if width == 8 else nodes[:, 24:28]
    interior = 0
    for w in words.ravel():
        if w == 0xFFFFFFFF:
            continue
        if w & 0x80000000:
            first, count = int(w & 0x0FFFFFFF), int((w >> 28) & 7) + 1
            seen[first:first + count] += 1
        else:
            interior += 1
    assert (seen == 1).all() and interior == nodes.shape[0] - 1
    oracle.set_device_tree(tree)
    try:
        p4, t4 = oracle.trace_primary(r.desc, mode=4)
    finally:
        oracle.set_device_tree(None)
    p0, t0 = oracle.trace_primary(r.desc, mode=0)
    assert np.array_equal(p4, p0) and np.array_equal(t4, t0)
    r.close()


def test_errors_are_status_codes():
    import ctypes as C
    from v4ray_hip import _abi, V4RayError
    lib = _abi.device_lib()
    h = C.c_void_p()
    assert lib.v4r_renderer_create(None, 0, C.byref(h)) == -1
    assert b"NULL" in lib.v4r_last_error()
    scene = scenes.load("c3", 32, 32)
    desc = scene.flatten(scene.renderer_param, scene.camera_param)
    bad = _abi.SceneDesc.from_buffer_copy(desc.contents)
    bad.n_nodes -= 1
    assert lib.v4r_renderer_create(C.byref(bad), 0, C.byref(h)) == -1
    assert lib.v4r_renderer_create(desc, 99, C.byref(h)) == -2
