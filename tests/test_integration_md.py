"""INTEGRATION.md's Rust binding against include/*.h (CPU tier).

There is no Rust toolchain in the image, so the `extern "C"` text a maintainer of the reference would paste next to
src/adsb.rs:92 has never been compiled.  What CAN be checked is drift: every `pub fn adsb_*` of the markdown must exist in
the headers with the same number of parameters, every `#[repr(C)]` struct must list the C struct's fields by the same
names in the same order (with integer / pointer types of the same width), and every `pub const ADSB_*` must carry the
header's value."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rust_blocks():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return "\n".join(re.findall(r"```rust\n(.*?)```", md, flags=re.S))


def _headers():
    out = ""
    for name in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if name.endswith(".h"):
            out += open(os.path.join(ROOT, "include", name)).read() + "\n"
    out = re.sub(r"/\*.*?\*/", " ", out, flags=re.S)    # comments
    return re.sub(r"//[^\n]*", " ", out)


def _split_args(s):
    """top-level comma split (function pointer types do not occur in this ABI)"""
    s = s.strip()
    if not s or s == "void":
        return []
    return [a.strip() for a in s.split(",")]


def _snake(name):
    return re.sub(r"(?<!^)(?=[A-Z])", "_", name).lower()


RUST_WIDTH = {"u8": 1, "i8": 1, "u16": 2, "i16": 2, "u32": 4, "i32": 4, "c_int": 4, "u64": 8, "i64": 8, "usize": 8, "f64": 8, "f32": 4}
C_WIDTH = {"uint8_t": 1, "int8_t": 1, "char": 1, "uint16_t": 2, "int16_t": 2, "uint32_t": 4, "int32_t": 4, "int": 4,
           "uint64_t": 8, "int64_t": 8, "size_t": 8, "double": 8, "float": 4}


def _c_structs(h):
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", h, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # `type a, b` declares several fields of one type
            tm = re.match(r"(?:const\s+)?([\w\s]+?)\s*(\*?)\s*(\w+\s*(?:\[\w+\])?(?:\s*,\s*\*?\s*\w+\s*(?:\[\w+\])?)*)$", decl)
            assert tm, decl
            ctype, ptr, names = tm.groups()
            for k, item in enumerate(names.split(",")):
                im = re.match(r"\s*(\*?)\s*(\w+)\s*(?:\[(\w+)\])?\s*$", item)
                assert im, decl
                is_ptr = bool(ptr) if k == 0 else bool(im.group(1))
                arr = im.group(3)
                width = 8 if is_ptr else C_WIDTH[ctype.split()[-1]]
                fields.append((im.group(2), width, int(arr) if arr and arr.isdigit() else (arr or None)))
        out[m.group(3)] = fields
    return out


def _rust_structs(r):
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[derive\([^\)]*\)\])?\s*pub struct (\w+)\s*\{(.*?)\}", r, flags=re.S):
        body = re.sub(r"//[^\n]*", " ", m.group(2))
        fields = []
        for decl in body.split(","):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"(?:pub\s+)?(\w+)\s*:\s*(.+)$", decl)
            assert fm, decl
            fname, rtype = fm.groups()
            am = re.match(r"\[(\w+);\s*(\w+)\]", rtype)
            if am:
                fields.append((fname, RUST_WIDTH[am.group(1)], int(am.group(2))))
            elif rtype.startswith("*"):
                fields.append((fname, 8, None))
            else:
                fields.append((fname, RUST_WIDTH[rtype.split("::")[-1]], None))
        out[m.group(1)] = fields
    return out


def test_every_rust_fn_matches_a_header_prototype():
    r, h = _rust_blocks(), _headers()
    rust_fns = {m.group(1): _split_args(m.group(2)) for m in re.finditer(r"pub fn (adsb_\w+)\s*\((.*?)\)\s*(?:->[^;]*)?;", r, flags=re.S)}
    assert len(rust_fns) >= 15, sorted(rust_fns)
    c_fns = {m.group(1): _split_args(m.group(2)) for m in re.finditer(r"\b(adsb_\w+)\s*\(([^;{}]*?)\)\s*;", h, flags=re.S)}
    for name, args in rust_fns.items():
        assert name in c_fns, f"INTEGRATION.md declares {name}, include/*.h does not"
        assert len(args) == len(c_fns[name]), (name, args, c_fns[name])
        for ra, ca in zip(args, c_fns[name]):     # pointer-ness must agree, parameter by parameter
            assert (("*" in ra.split(":", 1)[1]) == ("*" in ca)), (name, ra, ca)


def test_every_repr_c_struct_matches_the_c_struct():
    rs, cs = _rust_structs(_rust_blocks()), _c_structs(_headers())
    checked = 0
    for rname, rfields in rs.items():
        if [f[0] for f in rfields] == ["_private"]:
            continue                                   # opaque handle
        cname = _snake(rname)
        assert cname in cs, f"{rname} -> {cname}: no such struct in include/*.h"
        cf = cs[cname]
        assert [f[0] for f in rfields] == [f[0] for f in cf], (rname, [f[0] for f in rfields], [f[0] for f in cf])
        for (fn, rw, ra), (_, cw, ca) in zip(rfields, cf):
            assert rw == cw, (rname, fn, rw, cw)
            assert (ra or None) == (ca or None), (rname, fn, ra, ca)
        checked += 1
    assert checked >= 5, sorted(rs)


def test_rust_constants_carry_the_header_values():
    r, h = _rust_blocks(), _headers()
    consts = {m.group(1): m.group(2) for m in re.finditer(r"pub const (ADSB_\w+)\s*:\s*[\w:]+\s*=\s*(-?\w+)\s*;", r)}
    assert len(consts) >= 5
    for name, val in consts.items():
        m = re.search(r"#define\s+" + name + r"\s+\(?\s*(-?\w+)\s*\)?", h) or re.search(r"\b" + name + r"\s*=\s*(-?\w+)", h)
        assert m, f"{name} is not defined in include/*.h"
        assert int(val, 0) == int(m.group(1).rstrip("uUlL"), 0), (name, val, m.group(1))
