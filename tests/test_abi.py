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


# ---- the three descriptions of the ABI agree: the C header, the ctypes mirror, the Rust binding text of the integration notes ----------
CTYPES_OF = {"v4r_bvh_node": _abi.BvhNode, "v4r_prim": _abi.Prim, "v4r_xform": _abi.Xform, "v4r_tri_shading": _abi.TriShading,
             "v4r_medium": _abi.Medium, "v4r_material": _abi.Material, "v4r_texture": _abi.Texture, "v4r_image": _abi.Image,
             "v4r_perlin": _abi.Perlin, "v4r_camera": _abi.Camera, "v4r_scene_desc": _abi.SceneDesc, "v4r_stats": _abi.Stats,
             "v4r_tree_info": _abi.TreeInfo}
RUST_SCALAR = {"f32": (4, 4), "f64": (8, 8), "u32": (4, 4), "i32": (4, 4), "u64": (8, 8), "u8": (1, 1)}
C_TO_RUST = {"float": "f32", "double": "f64", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "int": "i32",
             "void": "std::ffi::c_void", "char": "std::os::raw::c_char"}


def header_structs():
    """[(struct name, [field names in order])] of include/v4ray_hip.h."""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "v4ray_hip.h")).read(), flags=re.S)
    out = []
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", text, flags=re.S):
        names = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if decl:
                names += [re.match(r"[\w\s]*?\**\s*(\w+)(?:\[\d+\])*$", d.strip()).group(1) for d in decl.split(",")]
        out.append((m.group(1), names))
    return out


def c_layout(tmp_path):
    """{struct: (size, {field: offset})} as gcc lays the header's structs out."""
    src = ['#include <stddef.h>', '#include <stdio.h>', '#include "v4ray_hip.h"', "int main(void) {"]
    for name, fields in header_structs():
        src.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f in fields:
            src.append(f'printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    src.append("return 0; }")
    c_file, exe = tmp_path / "layout.c", tmp_path / "layout"
    c_file.write_text("\n".join(src))
    import subprocess
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(c_file)], check=True)
    out = subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE, text=True).stdout
    layout = {}
    for line in out.splitlines():
        key, value = line.split()
        if "." in key:
            s, f = key.split(".")
            layout[s][1][f] = int(value)
        else:
            layout[key] = (int(value), {})
    return layout


def rust_text():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```rust\n(.*?)```", doc, flags=re.S)
    assert blocks, "INTEGRATION.md holds no Rust block"
    return "\n".join(blocks)


def rust_structs(text):
    """{Rust struct: [(field, type text)]}"""
    return {m.group(1): [(f.group(1), f.group(2).strip()) for f in re.finditer(r"pub (\w+): ([^,\n]+),", m.group(2))]
            for m in re.finditer(r"#\[repr\(C\)\][^\n]*\npub struct (\w+) \{(.*?)\n\}", text, flags=re.S)}


def rust_layout(structs):
    """repr(C) layout rules applied to the parsed Rust structs: {struct: (size, align, {field: offset})}."""
    done = {}

    def size_align(t):
        t = t.strip()
        if t.startswith("*"):
            return 8, 8
        m = re.match(r"\[(.*); (\d+)\]$", t)
        if m:
            s, a = size_align(m.group(1))
            return s * int(m.group(2)), a
        if t in RUST_SCALAR:
            return RUST_SCALAR[t]
        return lay(t)[:2]

    def lay(name):
        if name in done:
            return done[name]
        off, align, offsets = 0, 1, {}
        for field, t in structs[name]:
            s, a = size_align(t)
            off = (off + a - 1) // a * a
            offsets[field] = off
            off += s
            align = max(align, a)
        done[name] = ((off + align - 1) // align * align, align, offsets)
        return done[name]

    return {name: lay(name) for name in structs}


def camel(name):
    return "".join(p.capitalize() for p in name.split("_"))


def test_ctypes_and_rust_struct_layouts_match_the_header(tmp_path):
    """Field order, offsets and sizes of every struct of include/v4ray_hip.h as gcc lays them out, against the ctypes
    mirror the tests call through and against the #[repr(C)] structs of the Rust binding text in the integration notes."""
    layout = c_layout(tmp_path)
    rust = rust_structs(rust_text())
    rl = rust_layout(rust)
    assert set(CTYPES_OF) == set(layout), sorted(set(layout) ^ set(CTYPES_OF))
    for name, fields in header_structs():
        size, offsets = layout[name]
        ct = CTYPES_OF[name]
        assert C.sizeof(ct) == size, name
        assert [f for f, _ in ct._fields_] == fields, name
        for f in fields:
            assert getattr(ct, f).offset == offsets[f], (name, f)
        rname = camel(name)
        assert rname in rust, f"INTEGRATION.md has no #[repr(C)] struct {rname}"
        rfields = [("type" if f == "ty" else f) for f, _ in rust[rname]]
        assert rfields == fields, (rname, rfields, fields)
        rsize, _, roffsets = rl[rname]
        assert rsize == size, (rname, rsize, size)
        for f in fields:
            assert roffsets["ty" if f == "type" else f] == offsets[f], (rname, f)


def test_rust_externs_match_the_header():
    """Every entry point of include/v4ray_hip.h is declared in the Rust text with the same argument names, order and types."""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "v4ray_hip.h")).read(), flags=re.S)

    def to_rust(c_type):
        m = re.match(r"(const\s+)?(\w+)\s*((?:\*\s*)*)$", c_type.strip())
        t = C_TO_RUST.get(m.group(2)) or camel(m.group(2))
        for k in range(m.group(3).count("*")):
            t = ("*const " if m.group(1) and k == 0 else "*mut ") + t
        return t

    want = {}
    for m in re.finditer(r"^\s*((?:const\s+)?\w+\s*\**)\s*(v4r_\w+)\(([^)]*)\);", text, flags=re.M):
        args = []
        if " ".join(m.group(3).split()) != "void":
            for a in " ".join(m.group(3).split()).split(","):
                am = re.match(r"(.*?)(\w+)((?:\[\d*\])*)$", a.strip())
                args.append((am.group(2), to_rust(am.group(1) + (" *" if am.group(3) else ""))))
        want[m.group(2)] = (args, None if m.group(1).strip() == "void" else to_rust(m.group(1)))
    assert sorted(want) == sorted(_abi.DEVICE_SYMBOLS)
    have = {}
    for m in re.finditer(r"pub fn (v4r_\w+)\(([^)]*)\)(?: -> ([^;]+))?;", rust_text()):
        args = [tuple(x.strip() for x in a.split(":", 1)) for a in m.group(2).split(",") if a.strip()]
        have[m.group(1)] = (args, m.group(3).strip() if m.group(3) else None)
    assert sorted(have) == sorted(want), sorted(set(have) ^ set(want))
    for name in want:
        assert have[name] == want[name], (name, have[name], want[name])


def test_rust_constants_match_the_header():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "v4ray_hip.h")).read(), flags=re.S)
    rust = dict((m.group(1), m.group(2)) for m in re.finditer(r"pub const (V4R_\w+): \w+ = ([^;]+);", rust_text()))
    checked = 0
    for m in re.finditer(r"#define (V4R_\w+) \(?(-?(?:0x[0-9A-Fa-f]+|\d+))u?\)?\s*$", text, flags=re.M):
        assert int(rust[m.group(1)], 0) == int(m.group(2), 0), m.group(1)
        checked += 1
    for m in re.finditer(r"enum \{(.*?)\};", text, flags=re.S):
        for e in m.group(1).split(","):
            if e.strip():
                name, value = [x.strip() for x in e.split("=")]
                assert int(rust[name], 0) == int(value, 0), name
                checked += 1
    assert checked >= 30
