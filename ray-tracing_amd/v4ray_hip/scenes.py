"""The benchmark / parity scenes, as dictionaries in the reference's CLI JSON schema (src/json.rs:234-250).

c1  data/scene.json verbatim (Cornell walls + bunny + cuboid + triangle + fog sphere + three spheres)
c2  "Ray Tracing in One Weekend" random-spheres scene, generated from a seeded PCG32 stream following the
    recipe of the reference's data/scene1_gen.py:68-180 (which is unseeded and writes the GUI format)
c3  standard Cornell box with two rotated cuboids and an `important` ceiling light
c4  bunny.obj midpoint-subdivided twice (79 488 triangles) + earth-textured sphere inside the c1 walls
c5  data/zy_scene.json verbatim, resolution overridden
"""
import copy
import json
import math
import os

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "assets")


class _Pcg32:
    def __init__(self, seed, seq=0xda3e39cb94b95bdb):
        self.state, self.inc = 0, ((seq << 1) | 1) & 0xFFFFFFFFFFFFFFFF
        self.next()
        self.state = (self.state + seed) & 0xFFFFFFFFFFFFFFFF
        self.next()

    def next(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & 0xFFFFFFFFFFFFFFFF
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & 0xFFFFFFFF

    def uniform(self):
        return (self.next() >> 8) / 16777216.0


def _load(name):
    with open(os.path.join(ASSETS, name)) as f:
        return json.load(f)


def c1(width=800, height=800):
    s = _load("scene.json")
    s["renderer"]["width"], s["renderer"]["height"] = width, height
    return s


def c2(width=1920, height=1080, seed=20210625):
    rng = _Pcg32(seed)
    u = rng.uniform
    objects = [{
        "shape": {"type": "sphere", "center": [0.0, -1000.0, 0.0], "radius": 1000.0},
        "material": {"type": "lambertian", "texture": {
            "type": "checker", "density": 10.0,
            "odd": {"type": "solid-color", "color": [0.2, 0.3, 0.1]},
            "even": {"type": "solid-color", "color": [0.9, 0.9, 0.9]}}},
    }]
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose_mat = u()
            center = [a + 0.9 * u(), 0.2, b + 0.9 * u()]
            if math.sqrt((center[0] - 4.0) ** 2 + (center[1] - 0.2) ** 2 + center[2] ** 2) < 0.9:
                continue
            if choose_mat < 0.8:
                mat = {"type": "lambertian", "texture": {"type": "solid-color", "color": [u() * u(), u() * u(), u() * u()]}}
            elif choose_mat < 0.95:
                mat = {"type": "metal", "albedo": [0.5 + 0.5 * u(), 0.5 + 0.5 * u(), 0.5 + 0.5 * u()], "fuzz": 0.5 * u()}
            else:
                mat = {"type": "dielectric", "ir": 1.5}
            objects.append({"shape": {"type": "sphere", "center": center, "radius": 0.2}, "material": mat})
    objects.append({"shape": {"type": "sphere", "center": [0.0, 1.0, 0.0], "radius": 1.0},
                    "material": {"type": "dielectric", "ir": 1.5}})
    objects.append({"shape": {"type": "sphere", "center": [-4.0, 1.0, 0.0], "radius": 1.0},
                    "material": {"type": "lambertian", "texture": {"type": "solid-color", "color": [0.4, 0.2, 0.1]}}})
    objects.append({"shape": {"type": "sphere", "center": [4.0, 1.0, 0.0], "radius": 1.0},
                    "material": {"type": "metal", "albedo": [0.7, 0.6, 0.5], "fuzz": 0.0}})
    return {
        "renderer": {"width": width, "height": height, "max_depth": 50},
        "camera": {"look_from": [13.0, 2.0, 3.0], "look_at": [0.0, 0.0, 0.0], "vfov": 20.0, "up": [0.0, 1.0, 0.0],
                   "aperture": 0.1, "focus_dist": 10.0},
        "background": [1.0, 1.0, 1.0],
        "objects": objects,
    }


_WHITE = {"name": "white", "type": "lambertian", "texture": {"type": "solid-color", "color": [0.73, 0.73, 0.73]}}
_RED = {"name": "red", "type": "lambertian", "texture": {"type": "solid-color", "color": [0.65, 0.05, 0.05]}}
_GREEN = {"name": "green", "type": "lambertian", "texture": {"type": "solid-color", "color": [0.12, 0.45, 0.15]}}


def _cornell_walls():
    return [
        {"shape": {"type": "xy-rect", "x0": 0, "x1": 555, "y0": 0, "y1": 555, "z": 555, "positive": False}, "material": "white"},
        {"shape": {"type": "zx-rect", "z0": 0, "z1": 555, "x0": 0, "x1": 555, "y": 0, "positive": True}, "material": "white"},
        {"shape": {"type": "zx-rect", "z0": 0, "z1": 555, "x0": 0, "x1": 555, "y": 555, "positive": False}, "material": "white"},
        {"shape": {"type": "yz-rect", "y0": 0, "y1": 555, "z0": 0, "z1": 555, "x": 555, "positive": False}, "material": "green"},
        {"shape": {"type": "yz-rect", "y0": 0, "y1": 555, "z0": 0, "z1": 555, "x": 0, "positive": True}, "material": "red"},
    ]


def _rot_y(deg):
    c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
    return [[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]]


_CORNELL_CAMERA = {"look_from": [278, 278, -800], "look_at": [278, 278, 0], "vfov": 40, "aperture": 0}


def c3(width=1024, height=1024):
    objects = _cornell_walls()
    objects.append({"important": True,
                    "shape": {"type": "zx-rect", "z0": 213, "z1": 343, "x0": 213, "x1": 343, "y": 554, "positive": False},
                    "material": {"type": "diffuse-light", "emit": {"type": "solid-color", "color": [15, 15, 15]}}})
    objects.append({"shape": {"type": "cuboid", "p0": [0, 0, 0], "p1": [165, 330, 165], "transform": _rot_y(15),
                              "translate": [265, 0, 295]}, "material": "white"})
    objects.append({"shape": {"type": "cuboid", "p0": [0, 0, 0], "p1": [165, 165, 165], "transform": _rot_y(-18),
                              "translate": [130, 0, 65]}, "material": "white"})
    return {
        "renderer": {"width": width, "height": height, "max_depth": 50},
        "camera": dict(_CORNELL_CAMERA),
        "objects": objects,
        "materials": [copy.deepcopy(_WHITE), copy.deepcopy(_RED), copy.deepcopy(_GREEN)],
    }


def c4(width=1920, height=1080, subdivide=2):
    objects = [{"shape": {"type": "mesh", "file": "bunny.obj", "subdivide": subdivide,
                          "transform": [[1000, 0, 0], [0, 1000, 0], [0, 0, 1000]], "translate": [300, 0, 200]},
                "material": "white"}]
    objects += _cornell_walls()
    objects.append({"important": True,
                    "shape": {"type": "zx-rect", "z0": 100, "z1": 450, "x0": 100, "x1": 450, "y": 554, "positive": False},
                    "material": {"type": "diffuse-light", "emit": {"type": "solid-color", "color": [7, 7, 7]}}})
    objects.append({"shape": {"type": "sphere", "center": [100, 290, 290], "radius": 90},
                    "material": {"type": "lambertian", "texture": {"type": "image", "file": "earthmap.jpg"}}})
    return {
        "renderer": {"width": width, "height": height, "max_depth": 50},
        "camera": dict(_CORNELL_CAMERA),
        "objects": objects,
        "materials": [copy.deepcopy(_WHITE), copy.deepcopy(_RED), copy.deepcopy(_GREEN)],
    }


def c5(width=3840, height=2160):
    s = _load("zy_scene.json")
    s["renderer"]["width"], s["renderer"]["height"] = width, height
    return s


CONFIGS = {"c1": c1, "c2": c2, "c3": c3, "c4": c4, "c5": c5}
# (width, height, samples per pixel) each configuration is quoted at
FULL_SIZE = {"c1": (800, 800, 64), "c2": (1920, 1080, 256), "c3": (1024, 1024, 1024), "c4": (1920, 1080, 512),
             "c5": (3840, 2160, 4096)}


def load(name, width=None, height=None, perlin_seed=1, pretransform_triangles=False, flags=0, **kw):
    """Build the named scene -> v4ray_hip.Scene (with renderer_param / camera_param filled in)."""
    from . import Scene
    fw, fh, _ = FULL_SIZE[name]
    d = CONFIGS[name](width or fw, height or fh, **kw)
    scene = Scene.from_json_text(json.dumps(d), ASSETS, perlin_seed)
    scene.set_options(pretransform_triangles, flags)
    return scene
