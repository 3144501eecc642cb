#!/usr/bin/env python3
"""Print the Rust side of include/v4ray_hip.h: one #[repr(C)] struct per C struct, the constants, and the extern "C"
block -- the binding text a maintainer of the reference would add (tests/test_abi.py checks the copy in the integration notes
against the header field by field).

    python tools/gen_rust_binding.py > /tmp/hip.rs
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALARS = {"float": "f32", "double": "f64", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "int": "i32", "int32_t": "i32",
           "void": "std::ffi::c_void", "char": "std::os::raw::c_char", "size_t": "usize"}
RENAME = {"type": "ty"}  # Rust keywords


def camel(name):
    return "".join(p.capitalize() for p in name.split("_"))


def rust_type(c_type):
    c_type = c_type.strip()
    m = re.match(r"(const\s+)?(\w+)\s*((?:\*\s*)*)$", c_type)
    const, base, stars = bool(m.group(1)), m.group(2), m.group(3).count("*")
    t = SCALARS.get(base) or camel(base)
    for k in range(stars):
        t = ("*const " if const and k == 0 else "*mut ") + t
    return t


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def structs(text):
    """(struct name, [(field, C type, array dimensions)]) of every `typedef struct name { ... } name;`."""
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # `base declarator, declarator, ...`; a '*' belongs to the declarator it stands before
            head = re.match(r"((?:const\s+)?\w+)\s*(.*)$", decl)
            base = head.group(1)
            for d in head.group(2).split(","):
                dm = re.match(r"(\**)\s*(\w+)((?:\[\d+\])*)$", d.strip())
                stars, name, dims = dm.group(1), dm.group(2), [int(x) for x in re.findall(r"\[(\d+)\]", dm.group(3))]
                fields.append((name, (base + " " + stars).strip(), dims))
        yield m.group(1), fields


def functions(text):
    for m in re.finditer(r"^\s*((?:const\s+)?\w+\s*\**)\s*(v4r_\w+)\(([^)]*)\);", text, flags=re.M):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        parsed = []
        if args != "void":
            for a in args.split(","):
                am = re.match(r"(.*?)(\w+)((?:\[\d*\])*)$", a.strip())
                c_type, arg, dims = am.group(1).strip(), am.group(2), am.group(3)
                if dims:  # an array parameter is a pointer
                    c_type += " *"
                parsed.append((arg, c_type))
        yield name, ret, parsed


def main():
    text = strip_comments(open(os.path.join(ROOT, "include", "v4ray_hip.h")).read())
    out = ["// Rust side of include/v4ray_hip.h (tools/gen_rust_binding.py).  build.rs: println!(\"cargo:rustc-link-lib=dylib=v4ray_hip\");",
           "#![allow(non_camel_case_types, dead_code)]"]
    for m in re.finditer(r"#define (V4R_\w+) \(?(-?\w+)\)?\s*$", text, flags=re.M):
        name, value = m.group(1), m.group(2)
        if name in ("V4RAY_HIP_H",):
            continue
        if value.startswith("-"):
            out.append(f"pub const {name}: i32 = {value};")
        elif re.fullmatch(r"0x[0-9A-Fa-f]+u|\d+u", value):
            out.append(f"pub const {name}: u32 = {value[:-1]};")
        elif re.fullmatch(r"\d+", value):
            out.append(f"pub const {name}: i32 = {value};")
    out.append("pub const V4R_NOALLOC_SCRATCH_BYTES: u32 = 48 << 20;")
    for m in re.finditer(r"enum \{(.*?)\};", text, flags=re.S):
        for e in m.group(1).split(","):
            e = " ".join(e.split())
            if e:
                name, value = [x.strip() for x in e.split("=")]
                out.append(f"pub const {name}: u32 = {value};")
    for name, fields in structs(text):
        out.append("#[repr(C)] #[derive(Clone, Copy)]")
        out.append(f"pub struct {camel(name)} {{")
        for fname, c_type, dims in fields:
            t = rust_type(c_type)
            for d in reversed(dims):
                t = f"[{t}; {d}]"
            out.append(f"    pub {RENAME.get(fname, fname)}: {t},")
        out.append("}")
    out.append("#[repr(C)] pub struct V4rRenderer { _private: [u8; 0] }")
    out.append('extern "C" {')
    for name, ret, args in functions(text):
        a = ", ".join(f"{RENAME.get(n, n)}: {rust_type(t)}" for n, t in args)
        r = "" if ret == "void" else f" -> {rust_type(ret)}"
        out.append(f"    pub fn {name}({a}){r};")
    out.append("}")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
