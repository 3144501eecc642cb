"""Host-side scene pipeline: JSON schema loader, shape expanders, bounding boxes, BVH::build, Camera::new,
OBJ / JPEG readers and the flattened tables (CPU only)."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from v4ray_hip import (Cuboid, Lambertian, Mesh, PerspectiveCameraParam, RendererParam, RenderResult, Scene,
                       SolidColor, Sphere, Triangle, V4RayError, XYRect, ZXRect, _abi, load_image, scenes)

ASSETS = scenes.ASSETS
GREY = Lambertian(SolidColor((0.5, 0.5, 0.5)))
PARAM = RendererParam(16, 8)
CAM = PerspectiveCameraParam((0, 0, -5), (0, 0, 0), 40.0)


def build_bvh(boxes):
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
    n = boxes.shape[0]
    nodes = (_abi.HostBvhNode * max(2 * n, 1))()
    count = C.c_uint32()
    _abi.check_host(_abi.host_lib().v4rh_bvh_build(boxes.ctypes.data_as(C.POINTER(C.c_float)), n, nodes, C.byref(count)))
    return [nodes[i] for i in range(count.value)]


# ---- BVH::build, bvh/bvh.rs:34-180 --------------------------------------------------------------------
def sphere_box(c, r):
    return [c[0] - r, c[1] - r, c[2] - r, c[0] + r, c[1] + r, c[2] + r]


def test_bvh_five_sphere_hand_case():
    """Five unit spheres on the x axis at 0, 2, 4, 10, 20.  Centroid extent is widest in x (20); buckets of
    width 20/5.99: centres 0,2 -> bucket 0, 4 -> 1, 10 -> 2 (2.995), 20 -> 5.  cost(i) = nL*area(L) + nR*area(R)
    with area = 2*|size|^2 (aabb.rs:63-65): i=0: L={0,2} size (4,2,2) area 48; R={4,10,20} size (18,2,2) area 664
    -> 96 + 1992 = 2088; i=1: L={0,2,4} (6,2,2) 88 -> 264, R={10,20} (12,2,2) 304 -> 608: 872; i=2..4: L={0,2,4,10}
    (12,2,2) 304 -> 1216, R={20} (2,2,2) 24 -> 1240.  Minimum at i=1 -> split {0,1,2} | {3,4}."""
    centres = [(0, 0, 0), (2, 0, 0), (4, 0, 0), (10, 0, 0), (20, 0, 0)]
    nodes = build_bvh([sphere_box(c, 1.0) for c in centres])
    assert len(nodes) == 9
    root = nodes[0]
    assert not root.leaf and root.depth == 0
    assert list(root.children_aabb[0]) == [-1, -1, -1, 5, 1, 1]
    assert list(root.children_aabb[1]) == [9, -1, -1, 21, 1, 1]
    # DFS pre-order: left subtree directly after the root (bvh.rs:60-61,138-139)
    left = nodes[root.children_index[0]]
    assert root.children_index[0] == 1 and not left.leaf and left.parent_index == 0 and left.depth == 1
    # left = {0,2,4}: extent 4, buckets 0, 2 (2.995), 5 -> i=0: 1*24 + 2*(4,2,2)=48 -> 120; i=2: 2*48 + 24 = 120; first minimum wins
    assert nodes[left.children_index[0]].leaf and nodes[left.children_index[0]].shape_index == 0
    inner = nodes[left.children_index[1]]
    assert not inner.leaf and list(inner.children_aabb[0]) == [1, -1, -1, 3, 1, 1]
    assert [nodes[i].shape_index for i in inner.children_index] == [1, 2]
    right = nodes[root.children_index[1]]
    assert [nodes[i].shape_index for i in right.children_index] == [3, 4]
    leaves = sorted(n.shape_index for n in nodes if n.leaf)
    assert leaves == [0, 1, 2, 3, 4]


def test_bvh_coincident_centroids_split_in_half():
    # extent < EPSILON on every axis -> the index list is halved (bvh.rs:63-83)
    nodes = build_bvh([sphere_box((1, 1, 1), r) for r in (1, 2, 3, 4)])
    root = nodes[0]
    assert list(root.children_aabb[0]) == [-1, -1, -1, 3, 3, 3]     # join of shapes 0,1
    assert list(root.children_aabb[1]) == [-3, -3, -3, 5, 5, 5]     # join of shapes 2,3
    assert sorted(n.shape_index for n in nodes if n.leaf) == [0, 1, 2, 3]


def test_bvh_every_shape_reachable_once_and_boxes_contain_children():
    rng = np.random.default_rng(0)
    n = 500
    lo = rng.uniform(-100, 100, size=(n, 3))
    boxes = np.concatenate([lo, lo + rng.uniform(0.1, 20, size=(n, 3))], axis=1).astype(np.float32)
    nodes = build_bvh(boxes)
    assert len(nodes) == 2 * n - 1
    assert sorted(x.shape_index for x in nodes if x.leaf) == list(range(n))

    def bounds(i):
        node = nodes[i]
        if node.leaf:
            return boxes[node.shape_index]
        out = []
        for c in range(2):
            b = bounds(node.children_index[c])
            assert np.array_equal(np.float32(list(node.children_aabb[c])), b)   # stored box == exact join of the subtree
            out.append(b)
        return np.concatenate([np.minimum(out[0][:3], out[1][:3]), np.maximum(out[0][3:], out[1][3:])])
    bounds(0)
    assert build_bvh(np.zeros((0, 6))) == []
    single = build_bvh([[0, 0, 0, 1, 1, 1]])
    assert len(single) == 1 and single[0].leaf


def test_flattened_bvh_folds_leaves_into_child_words():
    s = Scene((0, 0, 0))
    for c in [(0, 0, 0), (2, 0, 0), (4, 0, 0), (10, 0, 0), (20, 0, 0)]:
        s.add(Sphere(c, 1.0), GREY)
    d = s.flatten(PARAM, CAM).contents
    assert d.n_prims == 5 and d.n_nodes == 4 and d.bvh_depth == 3
    root = d.nodes[0]
    assert root.child0 == 1 and root.child1 == 3
    assert d.nodes[1].child0 == (_abi.V4R_LEAF_BIT | 0) and d.nodes[1].child1 == 2
    assert d.nodes[2].child0 == (_abi.V4R_LEAF_BIT | 1) and d.nodes[2].child1 == (_abi.V4R_LEAF_BIT | 2)
    assert d.nodes[3].child0 == (_abi.V4R_LEAF_BIT | 3) and d.nodes[3].child1 == (_abi.V4R_LEAF_BIT | 4)
    one = Scene((0, 0, 0))
    one.add(Sphere((0, 0, 0), 1.0), GREY)
    d1 = one.flatten(PARAM, CAM).contents
    assert d1.n_nodes == 0 and d1.n_prims == 1 and d1.bvh_depth == 0


# ---- bounding boxes -----------------------------------------------------------------------------------
def test_bounding_boxes():
    eps = np.float32(0.001)
    s = Scene((0, 0, 0))
    s.add(Sphere((1, 2, 3), 0.5), GREY)
    s.add(XYRect(0, 1, 2, 3, 5.0), GREY)
    s.add(ZXRect(0, 1, 2, 3, 5.0), GREY)
    s.add(Triangle([(0, 0, 0), (1, 0, 0), (0, 1, 0)]), GREY)
    s.add(Sphere((0, 0, 0), 1.0).with_transform([[2, 0, 0], [0, 1, 0], [0, 0, 1]], (10, 0, 0)), GREY)
    b = s.bounding_box(0)
    assert b.min == (0.5, 1.5, 2.5) and b.max == (1.5, 2.5, 3.5)
    b = s.bounding_box(1)   # aa_rect.rs:34-39: thin axis padded by EPSILON
    assert np.allclose(b.min, (0, 2, 5 - eps)) and np.allclose(b.max, (1, 3, 5 + eps)) and b.min[2] == float(np.float32(5) - eps)
    b = s.bounding_box(2)   # aa_rect.rs:46-51: ZX rect is (z0,z1,x0,x1,y)
    assert np.allclose(b.min, (2, 5 - eps, 0)) and np.allclose(b.max, (3, 5 + eps, 1))
    b = s.bounding_box(3)   # triangle.rs:37-48: only exactly-flat axes are padded
    assert b.min == (0, 0, float(-eps)) and b.max == (1, 1, float(eps))
    b = s.bounding_box(4)   # transform.rs:37-56
    assert b.min == (8, -1, -1) and b.max == (12, 1, 1)


# ---- Camera::new, camera.rs:78-112 --------------------------------------------------------------------
def test_camera_setup():
    s = Scene((0, 0, 0))
    s.add(Sphere((0, 0, 0), 1.0), GREY)
    cam = s.flatten(RendererParam(200, 100), PerspectiveCameraParam((0, 0, -2), (0, 0, 0), 90.0)).contents.camera
    h = math.tan(math.radians(45))
    # w = normalize(from - at) = -z; u = up x w = (0,1,0) x (0,0,-1) = (-1,0,0); v = w x u = (0,1,0)
    assert np.allclose(cam.u[:], (-1, 0, 0)) and np.allclose(cam.v[:], (0, 1, 0))
    focus = 2.0   # |from - at| when focus_dist is absent
    assert np.allclose(cam.horizontal[:], (-focus * 2 * h * 2, 0, 0), rtol=1e-6)     # aspect = 200/100 (renderer.rs:85)
    assert np.allclose(cam.vertical[:], (0, focus * 2 * h, 0), rtol=1e-6)
    assert np.allclose(cam.lower_left_corner[:], (focus * 2 * h, -focus * h, 0), rtol=1e-6, atol=1e-6)
    assert cam.lens_radius == 0 and cam.time0 == 0 and cam.time1 == 0
    cam2 = s.flatten(RendererParam(200, 100), PerspectiveCameraParam((0, 0, -2), (0, 0, 0), 90.0, aspect_ratio=1.0,
                                                                      aperture=0.5, focus_dist=4.0, time0=1, time1=2)).contents.camera
    assert cam2.lens_radius == 0.25 and (cam2.time0, cam2.time1) == (1, 2)
    assert np.allclose(cam2.horizontal[:], (-4 * 2 * h, 0, 0), rtol=1e-6)


def test_renderer_param_defaults():
    s = Scene((0, 0, 0))
    s.add(Sphere((0, 0, 0), 1.0), GREY)
    d = s.flatten(RendererParam(4, 4), CAM).contents
    assert d.max_depth == 20 and d.antialias == 1          # renderer.rs:331, :103
    d = s.flatten(RendererParam(4, 4, max_depth=7, antialias=False), CAM).contents
    assert d.max_depth == 7 and d.antialias == 0
    with pytest.raises(ValueError):
        RendererParam(0, 4)


# ---- shape expanders ----------------------------------------------------------------------------------
def test_cuboid_expands_to_six_rects_in_reference_order():
    s = Scene((0, 0, 0))
    s.add(Cuboid((1, 2, 3), (4, 6, 8)), GREY)
    d = s.flatten(PARAM, CAM).contents
    assert d.n_prims == 6
    T = _abi
    expect = [(T.V4R_PRIM_XY_RECT, (1, 4, 2, 6, 3), False), (T.V4R_PRIM_XY_RECT, (1, 4, 2, 6, 8), True),
              (T.V4R_PRIM_YZ_RECT, (2, 6, 3, 8, 1), False), (T.V4R_PRIM_YZ_RECT, (2, 6, 3, 8, 4), True),
              (T.V4R_PRIM_ZX_RECT, (3, 8, 1, 4, 2), False), (T.V4R_PRIM_ZX_RECT, (3, 8, 1, 4, 6), True)]
    for i, (kind, f, positive) in enumerate(expect):   # hittables/cuboid.rs:35-60
        p = d.prims[i]
        assert p.type & T.V4R_PRIM_TYPE_MASK == kind and tuple(p.f[:5]) == f
        assert bool(p.type & T.V4R_PRIM_POSITIVE) == positive and p.xform == T.V4R_NO_XFORM


def test_obj_mesh_vertex_normals_and_subdivision(tmp_path):
    obj = tmp_path / "quad.obj"
    obj.write_text("# two triangles and a quad\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\n"
                   "f 1 2 3\nf 1 3 4\nf 1 5 2 3\n")
    s = Scene((0, 0, 0))
    s.add(Mesh(obj), GREY)
    d = s.flatten(PARAM, CAM).contents
    assert d.n_prims == 4          # the quad is fan-triangulated
    # vertex 4 (0,1,0) only touches the second triangle: its averaged normal is that face's normal +z (obj.rs:64-69,85-96)
    ts = d.tri_shading[d.prims[1].aux]
    assert np.allclose(ts.n[2][:], (0, 0, 1))
    assert tuple(ts.uv[0][:]) == (0.0, 0.0)
    tri0 = d.prims[0]
    assert tuple(tri0.f[:9]) == (0, 0, 0, 1, 0, 0, 1, 1, 0)       # v0, edge12, edge13 (triangle.rs:30-31)
    s2 = Scene((0, 0, 0))
    s2.add(Mesh(obj, subdivide=2), GREY)
    assert s2.flatten(PARAM, CAM).contents.n_prims == 4 * 16
    with pytest.raises(V4RayError, match="cannot open obj"):
        Scene((0, 0, 0)).add(Mesh(tmp_path / "missing.obj"), GREY)


def test_bunny_asset_loads():
    s = Scene((0, 0, 0))
    s.add(Mesh(os.path.join(ASSETS, "bunny.obj")), GREY)
    assert len(s) == 4968


def test_pretransform_bakes_mesh_transform():
    def build(bake):
        s = Scene((0, 0, 0))
        s.add(Triangle([(0, 0, 0), (1, 0, 0), (0, 1, 0)]).with_transform([[2, 0, 0], [0, 2, 0], [0, 0, 2]], (5, 0, 0)), GREY)
        s.add(Triangle([(0, 0, 0), (1, 0, 0), (0, 1, 0)]).with_transform([[-1, 0, 0], [0, 1, 0], [0, 0, 1]], (0, 0, 0)), GREY)
        s.set_options(bake, 0)
        return s, s.flatten(PARAM, CAM).contents
    s0, d0 = build(False)
    assert d0.n_xforms == 2 and d0.prims[0].xform == 0
    s1, d1 = build(True)
    assert d1.prims[0].xform == _abi.V4R_NO_XFORM and tuple(d1.prims[0].f[:9]) == (5, 0, 0, 2, 0, 0, 0, 2, 0)
    assert np.allclose(d1.tri_shading[d1.prims[0].aux].n[0][:], (0, 0, 2))   # M*n, not renormalised
    assert d1.prims[1].xform != _abi.V4R_NO_XFORM     # a mirroring transform flips front_face: kept as a transform


# ---- JSON schema, json.rs -----------------------------------------------------------------------------
def test_reference_scene_file_flattens():
    s = scenes.load("c1")
    d = s.flatten(s.renderer_param, s.camera_param).contents
    assert (d.width, d.height, d.max_depth) == (800, 800, 50)
    # 4968 bunny triangles + 6 cuboid rects + triangle + medium + 6 walls/light + 3 spheres (scene.json:21-223)
    assert d.n_prims == 4985 and d.n_nodes == 4984
    assert d.n_lights == 1 and d.prims[d.lights[0]].type & 0xFF == _abi.V4R_PRIM_ZX_RECT
    assert d.n_media == 1 and d.media[0].count == 1 and d.media[0].neg_inv_density == pytest.approx(-100.0)
    assert d.n_images == 1 and (d.images[0].width, d.images[0].height) == (1024, 512)
    assert d.n_perlin == 1 and d.n_xforms == 2      # bunny transform shared by all its triangles, cuboid transform
    assert s.camera_param.aperture == 0 and s.camera_param.focus_dist is None
    kinds = [d.materials[d.prim_material[i]].kind for i in range(4968, 4985)]
    assert kinds[6] == _abi.V4R_MAT_METAL and kinds[7] == _abi.V4R_MAT_ISOTROPIC
    zy = scenes.load("c5", 64, 36)
    dz = zy.flatten(zy.renderer_param, zy.camera_param).contents
    assert dz.n_prims == 15 and dz.n_perlin == 5 and dz.max_depth == 20


def test_json_visibility_importance_and_names():
    text = json.dumps({
        "renderer": {"width": 4, "height": 4}, "camera": {"look_from": [0, 0, -5], "look_at": [0, 0, 0], "vfov": 40},
        "background": [0.1, 0.2, 0.3],
        "textures": [{"name": "t", "type": "solid-color", "color": [1, 0, 0]}],
        "materials": [{"name": "m", "type": "lambertian", "texture": "t"},
                      {"name": "glow", "type": "diffuse-light", "emit": {"type": "solid-color", "color": [3, 3, 3]}}],
        "shapes": [{"name": "ball", "type": "sphere", "center": [0, 0, 0], "radius": 1}],
        "objects": [
            {"shape": "ball", "material": "m"},
            {"shape": "ball", "material": "m", "visible": False, "important": True},          # dropped (json.rs:687-699)
            {"shape": {"type": "sphere", "center": [0, 3, 0], "radius": 1, "translate": [1, 0, 0]}, "material": "glow", "important": True},
            {"shape": {"type": "constant-medium", "density": 0.5, "shape": "ball"}, "material": "m", "important": True},  # not samplable: dropped
        ]})
    s = Scene.from_json_text(text)
    d = s.flatten(s.renderer_param, s.camera_param).contents
    assert d.n_prims == 2 and d.n_lights == 1 and d.lights[0] == 1
    assert d.n_textures == 2 and d.n_materials == 2                      # named entries are built once
    assert d.prims[1].xform == 0 and tuple(d.xforms[0].fwd_t[:]) == (1, 0, 0) and tuple(d.xforms[0].inv_t[:]) == (-1, 0, 0)
    assert np.allclose(d.background[:], (0.1, 0.2, 0.3)) and tuple(d.environment[:]) == (0, 0, 0)


@pytest.mark.parametrize("mutate,message", [
    (lambda d: d["objects"].append({"shape": "nope", "material": "m"}), "unknown shape name"),
    (lambda d: d["objects"].append({"shape": {"type": "blob"}, "material": "m"}), "unknown shape type"),
    (lambda d: d["materials"].append({"name": "m2", "type": "lambertian", "texture": "missing"}) or
        d["objects"].append({"shape": "ball", "material": "m2"}), "unknown texture name"),
    (lambda d: d.pop("camera"), "missing field `camera`"),
    (lambda d: d["renderer"].update(width=-3), "unsigned integer"),
    (lambda d: d["textures"].append({"name": "loop", "type": "checker", "odd": "loop", "even": "t", "density": 1}) or
        d["materials"].append({"name": "m3", "type": "lambertian", "texture": "loop"}) or
        d["objects"].append({"shape": "ball", "material": "m3"}), "cycle"),
])
def test_json_errors_are_reported(mutate, message):
    d = {"renderer": {"width": 4, "height": 4}, "camera": {"look_from": [0, 0, -5], "look_at": [0, 0, 0], "vfov": 40},
         "textures": [{"name": "t", "type": "solid-color", "color": [1, 0, 0]}],
         "materials": [{"name": "m", "type": "lambertian", "texture": "t"}],
         "shapes": [{"name": "ball", "type": "sphere", "center": [0, 0, 0], "radius": 1}],
         "objects": [{"shape": "ball", "material": "m"}]}
    mutate(d)
    with pytest.raises(V4RayError, match=message):
        Scene.from_json_text(json.dumps(d))
    with pytest.raises(V4RayError, match="JSON"):
        Scene.from_json_text("{ not json")


# ---- image decode -------------------------------------------------------------------------------------
def test_jpeg_decoder_close_to_libjpeg():
    from PIL import Image
    ours = load_image(os.path.join(ASSETS, "earthmap.jpg"))
    ref = np.asarray(Image.open(os.path.join(ASSETS, "earthmap.jpg")).convert("RGB"))
    assert ours.shape == ref.shape == (512, 1024, 3)
    diff = np.abs(ours.astype(int) - ref.astype(int))
    assert diff.max() <= 3 and diff.mean() < 0.5      # different (both valid) IDCT roundings


def test_jpeg_420_and_ppm(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:37, 0:53]
    img = np.stack([xx * 4, yy * 6, (xx + yy) * 2], axis=2).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / "a.jpg", quality=95, subsampling=2)   # 4:2:0, odd size
    ours = load_image(tmp_path / "a.jpg")
    ref = np.asarray(Image.open(tmp_path / "a.jpg").convert("RGB"))
    assert ours.shape == ref.shape
    assert np.abs(ours.astype(int) - ref.astype(int)).mean() < 4    # replicated vs interpolated chroma
    Image.fromarray(img).save(tmp_path / "a.ppm")
    assert np.array_equal(load_image(tmp_path / "a.ppm"), img)
    with pytest.raises(V4RayError, match="failed to open image"):
        load_image(tmp_path / "none.jpg")
    (tmp_path / "bad.jpg").write_bytes(b"\xff\xd8\xff\xe0garbage")
    with pytest.raises(V4RayError, match="JPEG"):
        load_image(tmp_path / "bad.jpg")


def test_malformed_jpegs_are_errors_not_crashes(tmp_path):
    """Texture files are untrusted input: truncated files, table ids and bit counts out of range, and random byte
    damage in the headers or the entropy-coded data either decode to an image of the declared size or raise."""
    good = open(os.path.join(ASSETS, "earthmap.jpg"), "rb").read()

    def attempt(data, must_fail=False):
        (tmp_path / "x.jpg").write_bytes(bytes(data))
        try:
            img = load_image(tmp_path / "x.jpg")
        except V4RayError as e:
            assert "JPEG" in str(e) or "decode" in str(e)
            return False
        assert not must_fail
        assert img.ndim == 3 and img.shape[2] == 3 and img.dtype == np.uint8
        return True

    sos = good.index(b"\xff\xda")
    ns = good[sos + 4]
    assert ns == 3
    for k in range(ns):                       # Huffman table ids 4..15 in the scan header
        bad = bytearray(good)
        bad[sos + 6 + 2 * k] = 0xF0 if k else 0x4F
        attempt(bad, must_fail=True)
    bad = bytearray(good)                     # a scan header shorter than its component list
    bad[sos + 2:sos + 4] = (3).to_bytes(2, "big")
    attempt(bad, must_fail=True)
    dht = good.index(b"\xff\xc4")             # every DC symbol says "32 more bits"
    bad = bytearray(good)
    n_codes = sum(bad[dht + 5:dht + 21])
    bad[dht + 21:dht + 21 + n_codes] = bytes([32]) * n_codes
    attempt(bad, must_fail=True)
    attempt(good[:2] + b"\xff" * 50, must_fail=True)     # fill bytes up to the end of the file
    for cut in list(range(2, 700, 7)) + [len(good) // 2, len(good) - 1]:
        attempt(good[:cut], must_fail=cut < sos + 14)    # inside the headers; a cut scan decodes to a partly grey image
    rng = np.random.default_rng(11)
    decoded = 0
    for trial in range(120):
        bad = bytearray(good)
        region = sos + 14 if trial % 2 else len(bad)
        for _ in range(int(rng.integers(1, 6))):
            bad[int(rng.integers(2, region))] = int(rng.integers(0, 256))
        decoded += attempt(bad)
    assert decoded > 20                        # damage in the scan data usually still decodes (to a damaged image)


def test_written_images_round_trip(tmp_path):
    """The output format follows the file name (src/main.rs:70-72 saves through the `image` crate, which does the same):
    PNG, BMP, PPM and TGA files written here read back pixel for pixel; a format the writer lacks is an error."""
    from PIL import Image
    from v4ray_hip import write_image
    rng = np.random.default_rng(5)
    for shape in ((37, 53, 3), (1, 1, 3), (300, 301, 3)):   # 300 x 301 x 3 + filter bytes: more than one stored deflate block
        img = rng.integers(0, 256, size=shape, dtype=np.uint8)
        for ext in ("png", "bmp", "ppm", "tga", "PNG"):
            path = tmp_path / f"a.{ext}"
            write_image(path, img)
            back = np.asarray(Image.open(path).convert("RGB"))
            assert np.array_equal(back, img), (shape, ext)
    with pytest.raises(V4RayError, match="not supported"):
        write_image(tmp_path / "a.jpg", img)


# ---- RenderResult, renderer.rs:336-406 ----------------------------------------------------------------
def test_render_result_tonemap_table():
    res = RenderResult(4, 1)
    assert res.get_raw(0) is None
    a = np.zeros((1, 4, 3), np.float32)
    a[0, 0] = (0.0, 0.25, 1.0)
    a[0, 1] = (4.0, np.nan, 1e-9)
    a[0, 2] = (0.5, 0.5, 0.5)
    res.add(a)
    res.add(a)
    out, n = res.get_raw(0)
    assert n == 2 and res.get_raw(2) is None
    # u8(clamp(256*sqrt(sum/n), 0.5, 255.5)); NaN -> 0 (renderer.rs:378-402)
    assert tuple(out[0, 0]) == (0, 128, 255) and tuple(out[0, 1]) == (255, 0, 0)
    assert out[0, 2, 0] == int(256 * math.sqrt(0.5))


def test_scene_generators_are_deterministic():
    a, b = scenes.c2(64, 36), scenes.c2(64, 36)
    assert a == b and len(a["objects"]) > 400
    assert scenes.c2(64, 36, seed=1) != a
    s = scenes.load("c2", 64, 36)
    d = s.flatten(s.renderer_param, s.camera_param).contents
    assert d.n_prims == len(a["objects"]) and d.n_lights == 0
    assert d.camera.lens_radius == pytest.approx(0.05)
    c4 = scenes.load("c4", 64, 36, subdivide=1)
    assert c4.flatten(c4.renderer_param, c4.camera_param).contents.n_prims == 4968 * 4 + 7


def test_checkpoint_round_trip_and_corruption(tmp_path):
    """Accumulator checkpoint (sums + pass count + size + seed): exact round trip, previous file survives as
    the only file after a save, truncated / corrupted / foreign files are rejected with a message."""
    from v4ray_hip import V4RayError, load_checkpoint, save_checkpoint
    rng = np.random.default_rng(3)
    sums = rng.random((7, 5, 3), dtype=np.float32) * 100
    sums[0, 0, 0] = np.float32(np.inf)   # an overflowed pixel must survive too
    path = tmp_path / "job.ckpt"
    save_checkpoint(path, sums, 123, 0x5EED)
    got, n, seed = load_checkpoint(path)
    assert n == 123 and seed == 0x5EED and got.shape == (7, 5, 3) and np.array_equal(got, sums)
    assert not (tmp_path / "job.ckpt.tmp").exists()
    save_checkpoint(path, sums * 2, 246, 0x5EED)            # overwrite in place
    assert load_checkpoint(path)[1] == 246
    raw = path.read_bytes()
    assert len(raw) == 48 + 7 * 5 * 3 * 4 and raw[:8] == b"V4RCKPT1"
    (tmp_path / "cut.ckpt").write_bytes(raw[:-10])
    with pytest.raises(V4RayError, match="truncated"):
        load_checkpoint(tmp_path / "cut.ckpt")
    flipped = bytearray(raw)
    flipped[60] ^= 1
    (tmp_path / "bad.ckpt").write_bytes(bytes(flipped))
    with pytest.raises(V4RayError, match="checksum"):
        load_checkpoint(tmp_path / "bad.ckpt")
    (tmp_path / "other.ckpt").write_bytes(b"P6\n1 1\n255\n" + bytes(60))
    with pytest.raises(V4RayError, match="not a checkpoint"):
        load_checkpoint(tmp_path / "other.ckpt")
    with pytest.raises(V4RayError, match="cannot open"):
        load_checkpoint(tmp_path / "missing.ckpt")
