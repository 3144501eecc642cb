"""The Python package as a drop-in for the reference's `v4ray` extension module (src/py.rs:62-86, v4ray/*.pyi): the
reference's own front-end package (v4ray_frontend/, imported from /root/reference where it lies -- nothing of it is
copied here) must import against it and build one object of every type it registers.  Runs only where the reference
tree is present (the build container); the GPU box has no /root/reference."""
import importlib
import os
import sys
import uuid

import numpy as np
import pytest

import v4ray_hip

REFERENCE = "/root/reference"
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "v4ray_frontend")), reason="the reference tree is not present")


def test_module_surface_matches_the_stubs():
    """Names the stubs declare (v4ray/__init__.pyi, shape.pyi, material.pyi, texture.pyi), importable submodules included."""
    v4ray_hip.install_as("v4ray")
    v4ray = importlib.import_module("v4ray")
    for name in ("Scene", "AABB", "Ray", "HitRecord", "PerspectiveCameraParam", "RendererParam", "Renderer"):
        assert hasattr(v4ray, name), name
    from v4ray.shape import Sphere
    from v4ray.material import Dielectric, Lambertian, Metal
    from v4ray.texture import Checker, SolidColor
    assert importlib.import_module("v4ray.shape").Sphere is Sphere is v4ray_hip.Sphere
    assert importlib.import_module("v4ray_hip.material").Metal is Metal
    s = Sphere((1.0, 2.0, 3.0), 0.5)
    assert s.center == (1.0, 2.0, 3.0) and s.radius == 0.5
    box = s.bounding_box()
    assert box.min == (0.5, 1.5, 2.5) and box.max == (1.5, 2.5, 3.5)
    assert s.bounding_box(0.0, 1.0).max == box.max       # the two shutter times the Rust method takes (sphere.rs:162-165)
    Lambertian(Checker(SolidColor((1, 0, 0)), SolidColor((0, 1, 0)), 2.0)), Metal((0.5, 0.5, 0.5), 0.1), Dielectric(1.5)
    assert asyncio_iscoroutinefunction(v4ray.Renderer.render)


def asyncio_iscoroutinefunction(f):
    import asyncio
    return asyncio.iscoroutinefunction(f)


def test_sphere_hit_on_a_ray_packet():
    """shape.Sphere.hit(ray, t_min, t_max) -> HitRecord (v4ray/shape.pyi, sphere.rs:60-92,166-179) on eight rays: head-on,
    from inside, grazing miss, masked-off lane, hit behind t_max, hit before t_min."""
    from v4ray_hip import HitRecord, Ray, Sphere
    s = Sphere((0.0, 0.0, 5.0), 1.0)
    o = np.zeros((3, 8), np.float32)
    d = np.zeros((3, 8), np.float32)
    d[2] = 1.0
    o[:, 1] = (0, 0, 5)            # from the centre: the far root
    o[0, 2] = 1.5                  # passes beside the sphere
    o[0, 4] = 0.6                  # off-axis hit
    d[:, 5] = (0, 0, -1)           # looks away
    mask = np.ones(8, bool)
    mask[3] = False
    t_min = np.full(8, 1e-3, np.float32)
    t_max = np.full(8, np.inf, np.float32)
    t_max[6] = 3.0                 # both roots beyond t_max
    t_min[7] = 4.5                 # the near root before t_min: the far one counts
    h = s.hit(Ray(o, d * 3.0, mask=mask), t_min, t_max)   # directions are normalised on the way in
    assert isinstance(h, HitRecord)
    assert h.mask.tolist() == [True, True, False, False, True, False, False, True]
    assert np.allclose(h.t[[0, 1, 7]], [4.0, 1.0, 6.0])
    assert np.allclose(h.t[4], 5.0 - np.sqrt(1 - 0.36), rtol=1e-6)
    assert h.front_face.tolist() == [True, False, False, False, True, False, False, False]
    assert np.allclose(h.normal[:, 0], (0, 0, -1)) and np.allclose(h.normal[:, 1], (0, 0, -1)) and np.allclose(h.normal[:, 7], (0, 0, -1))
    assert np.allclose(h.p[:, 4], (0.6, 0, 4.2), atol=1e-6)
    # sphere_uv (sphere.rs:38-45) of the outward normal (0, 0, -1): theta = acos(0), phi = atan2(1, 0) + pi
    assert np.allclose(h.uv[:, 0], (0.75, 0.5), atol=1e-6)
    assert (h.t[~h.mask] == 0).all() and (h.p[:, ~h.mask] == 0).all()
    none = Sphere((0.0, 50.0, 0.0), 1.0).hit(Ray(o, d), t_min, t_max)   # no ray meets it: HitRecord::default
    assert not none.mask.any() and not none.t.any()
    with pytest.raises(ValueError):
        s.hit(Ray(o, d), np.zeros(4, np.float32), t_max)
    r = Ray(o, d * 2.0)
    assert r.origin.shape == (3, 8) and r.direction.shape == (3, 8) and r.time.shape == (8,) and r.mask.dtype == bool
    assert np.allclose(np.linalg.norm(r.direction, axis=0), 1.0)


def test_sphere_hit_agrees_with_the_oracle():
    """The packet method against the oracle's restatement of Sphere::hit, ray by ray."""
    import oracle
    from v4ray_hip import Lambertian, PerspectiveCameraParam, Ray, RendererParam, Scene, SolidColor, Sphere
    centre, radius = (0.3, -0.2, 4.0), 1.3
    scene = Scene((0, 0, 0))
    scene.add(Sphere(centre, radius), Lambertian(SolidColor((0.5, 0.5, 0.5))))
    desc = scene.flatten(RendererParam(8, 8), PerspectiveCameraParam((0, 0, 0), (0, 0, 1), 40.0))
    rng = np.random.default_rng(3)
    s = Sphere(centre, radius)
    for _ in range(16):
        o = rng.uniform(-1, 1, size=(3, 8)).astype(np.float32)
        d = (np.float32([[0], [0], [1]]) + rng.normal(size=(3, 8)) * 0.35).astype(np.float32)
        ray = Ray(o, d)
        h = s.hit(ray, np.full(8, 1e-3, np.float32), np.full(8, np.inf, np.float32))
        for k in range(8):
            ref = oracle.hit(desc, 0, ray.origin[:, k], ray.direction[:, k])
            assert bool(h.mask[k]) == ref["mask"]
            if ref["mask"]:
                assert np.allclose(h.t[k], ref["t"], rtol=2e-6) and np.allclose(h.p[:, k], ref["p"], atol=2e-6)
                assert np.allclose(h.normal[:, k], ref["normal"], atol=2e-6) and bool(h.front_face[k]) == ref["front_face"]
                assert np.allclose(h.uv[:, k], (ref["u"], ref["v"]), atol=2e-6)


@needs_reference
def test_reference_front_end_builds_its_objects_against_this_package():
    """`import v4ray_frontend` evaluates v4ray.Ray / v4ray.HitRecord / v4ray.AABB at import (v4ray_frontend/shape.py:11-13)
    and every registered type's apply() constructs through v4ray.shape / v4ray.material / v4ray.texture / v4ray
    (shape.py:74, material.py:72-159, texture.py:65,104, camera.py:81-103)."""
    v4ray_hip.install_as("v4ray")
    sys.path.insert(0, REFERENCE)
    try:
        for m in [k for k in sys.modules if k == "v4ray_frontend" or k.startswith("v4ray_frontend.")]:
            del sys.modules[m]
        fe = importlib.import_module("v4ray_frontend")
    finally:
        sys.path.remove(REFERENCE)
    assert [t.kind() for t in fe.shapes] == ["sphere"]
    red, green = uuid.uuid4(), uuid.uuid4()
    textures = {red: fe.SolidColor.apply([(255, 0, 0)], {}), green: fe.SolidColor.apply([(0, 255, 0)], {})}
    check = uuid.uuid4()
    textures[check] = fe.Checker.apply([red, green, 2.0], textures)
    assert isinstance(textures[red], v4ray_hip.SolidColor) and isinstance(textures[check], v4ray_hip.Checker)
    materials = []
    for mt in fe.materials:
        data = {"lambertian": [check], "metal": [(200, 200, 255), 0.1], "dielectric": [1.5]}[mt.kind()]
        materials.append(mt.apply(data, textures))
        mt.apply_preview(data, textures)
    assert [type(m) for m in materials] == [v4ray_hip.Lambertian, v4ray_hip.Metal, v4ray_hip.Dielectric]
    shapes = fe.Sphere.apply([0.0, 1.0, 0.0, 1.0])
    assert len(shapes) == 1 and shapes[0].radius == 1.0
    cam_data = [p.default if getattr(p, "default", None) is not None else 0.0 for p in fe.PerspectiveCamera.properties()]
    assert fe.PerspectiveCamera.validate(cam_data)
    camera = fe.PerspectiveCamera.apply(cam_data)
    fe.PerspectiveCamera.apply_preview(cam_data)
    assert isinstance(camera, v4ray_hip.PerspectiveCameraParam)
    # what main.py's State.generate does with them (main.py:1985): a scene the host library flattens for the device
    scene = v4ray_hip.Scene((0.7, 0.8, 1.0), (0.0, 0.0, 0.0))
    for m in materials:
        scene.add(shapes[0], m)
    desc = scene.flatten(v4ray_hip.RendererParam(64, 48), camera)
    assert desc.contents.n_prims == 3 and desc.contents.width == 64


def _bare_renderer(render_mean, render_pass=None):
    """A Renderer without a device behind it: only the bookkeeping of the awaited render() calls."""
    import threading
    r = v4ray_hip.Renderer.__new__(v4ray_hip.Renderer)
    r._pending, r._pending_lock, r._draining = [], threading.Lock(), False
    r._render_mean = render_mean
    r.render_pass = render_pass or (lambda: render_mean(1))
    r._h = None
    return r


def test_awaited_render_calls_are_served_in_batches():
    """PyRenderer.render (src/renderer.rs:462-476) as the front end uses it (main.py:1986-1990): calls waiting together are
    handed to one job; every call gets an answer, and the jobs' sizes add up to the number of calls."""
    import asyncio
    import time
    sizes = []

    def render_mean(n):
        sizes.append(n)
        time.sleep(0.01)   # calls made meanwhile wait for the next job
        return np.full((2, 3, 3), float(n), dtype=np.float32)

    r = _bare_renderer(render_mean)

    async def worker(k, out):
        for _ in range(5):
            out.append(await r.render(shared=True))

    async def run():
        out = []
        await asyncio.gather(*[worker(k, out) for k in range(8)])
        return out

    got = asyncio.run(run())
    assert len(got) == 40 and sum(sizes) == 40
    assert max(sizes) > 1            # calls did share jobs
    assert r._pending == [] and r._draining is False


def test_awaited_render_calls_get_distinct_passes_by_default():
    """Without shared=True every awaited call is a pass of its own (src/renderer.rs:450-476 returns one image per call),
    whatever else is in flight; shared and distinct calls may be mixed."""
    import asyncio
    served = []

    def render_pass():
        served.append(len(served))
        return np.full((2, 3, 3), float(served[-1]), dtype=np.float32)

    def render_mean(n):
        return np.full((2, 3, 3), -float(n), dtype=np.float32)

    r = _bare_renderer(render_mean, render_pass)

    async def run():
        return await asyncio.gather(*([r.render() for _ in range(6)] + [r.render(shared=True) for _ in range(3)]))

    got = asyncio.run(run())
    distinct = sorted(float(g[0, 0, 0]) for g in got[:6])
    assert distinct == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]      # six calls, six different passes
    assert all(float(g[0, 0, 0]) < 0 for g in got[6:])       # the shared ones came from batch jobs
    assert r._draining is False


def test_a_failing_drain_does_not_block_later_render_calls():
    """An exception that is not an Exception subclass (e.g. KeyboardInterrupt in the executor thread) must not leave the
    drain flag standing: later calls start a new drain."""
    import asyncio
    calls = []

    def render_mean(n):
        calls.append(n)
        if len(calls) == 1:
            raise KeyboardInterrupt()
        return np.zeros((2, 3, 3), dtype=np.float32)

    r = _bare_renderer(render_mean)

    async def one():
        return await asyncio.wait_for(r.render(shared=True), timeout=5)

    with pytest.raises(BaseException):
        asyncio.run(one())
    assert r._draining is False and r._pending == []
    assert asyncio.run(one()).shape == (2, 3, 3)


def test_awaited_render_calls_all_see_a_failure():
    import asyncio

    def render_mean(n):
        raise v4ray_hip.V4RayError("device lost")

    r = _bare_renderer(render_mean)

    async def run():
        return await asyncio.gather(*([r.render(shared=True) for _ in range(4)] + [r.render() for _ in range(2)]), return_exceptions=True)

    got = asyncio.run(run())
    assert len(got) == 6 and all(isinstance(e, v4ray_hip.V4RayError) for e in got)
    assert r._draining is False


def test_close_waits_for_the_render_in_flight():
    """close() takes the lock a render holds, so the handle is never destroyed under the thread that is using it."""
    import threading
    import time
    r = _bare_renderer(lambda n: None)
    r._lock = threading.Lock()
    order = []

    def rendering():
        with r._lock:
            order.append("render starts")
            time.sleep(0.05)
            order.append("render ends")

    t = threading.Thread(target=rendering)
    t.start()
    while not order:
        time.sleep(0.001)
    r.close()
    order.append("closed")
    t.join()
    assert order == ["render starts", "render ends", "closed"]


def test_the_package_defines_every_top_level_name_once():
    import ast
    import os
    for name in ("__init__.py", "_abi.py", "scenes.py", "distributed.py"):
        path = os.path.join(os.path.dirname(v4ray_hip.__file__), name)
        seen = {}
        for node in ast.parse(open(path).read()).body:
            if isinstance(node, (ast.ClassDef, ast.FunctionDef)):
                assert node.name not in seen, f"{name}: {node.name} is defined at lines {seen[node.name]} and {node.lineno}"
                seen[node.name] = node.lineno
            if isinstance(node, ast.ClassDef):
                methods = {}
                for m in node.body:
                    if isinstance(m, ast.FunctionDef):
                        decorated = any(isinstance(d, ast.Attribute) and d.attr in ("setter", "deleter") for d in m.decorator_list)
                        assert decorated or m.name not in methods, f"{name}: {node.name}.{m.name} is defined twice"
                        methods[m.name] = m.lineno
