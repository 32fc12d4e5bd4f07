#!/usr/bin/env python3
"""Generates rust/ffi.rs -- the `extern "C"` declarations a tfhe-rs maintainer would put in `tfhe/src/hip_backend/ffi.rs` --
from the C headers of the boundary (include/tfhe_hip.h, tfhe_hip_strings.h, tfhe_hip_experimental.h): what `bindgen` would
emit, written by a small C-declaration parser because neither bindgen nor rustc exists in this image (the output is
therefore UNCOMPILED; tests/test_rust_ffi.py checks header <=> ffi.rs <=> `nm -D libtfhe_hip.so` symbol by symbol and
argument count by argument count).  Conventions mirrored from the reference's own C API (tfhe/src/c_api/utils.rs:3-28,
c_api/shortint/server_key/pbs.rs:17-167, tfhe/cbindgen.toml): every function returns c_int (0 = ok), out-parameters are
`*mut *mut T`, opaque handles are zero-sized #[repr(C)] structs.

  python tools/gen_rust_ffi.py            # rewrites rust/ffi.rs
  python tools/gen_rust_ffi.py --check    # exit 1 if rust/ffi.rs is stale
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = ["tfhe_hip.h", "tfhe_hip_strings.h", "tfhe_hip_experimental.h"]
OUT = os.path.join(ROOT, "rust", "ffi.rs")

PRIM = {
    "int": "c_int", "unsigned": "c_uint", "unsigned int": "c_uint", "char": "c_char", "void": "c_void", "double": "f64", "float": "f32",
    "size_t": "usize", "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int8_t": "i8", "int16_t": "i16",
    "int32_t": "i32", "int64_t": "i64",
}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def camel(name):
    return "".join(p.capitalize() for p in name.split("_"))


def parse_header(path):
    """-> (opaque struct names, plain structs {name: [(ctype, field)]}, enum constants [(name, value)], functions [(name, ret, [(ctype, argname)])])"""
    text = strip_comments(open(path).read())
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)            # preprocessor lines
    text = text.replace('extern "C" {', " ").replace("extern \"C\"", " ")
    opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text)
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s+\w*\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ctype, names = decl.rsplit(" ", 1)[0], decl.rsplit(" ", 1)[1]
            rest = decl[len(ctype):]
            for nm in rest.split(","):
                fields.append((ctype.strip(), nm.strip()))
        structs[name] = fields
    enums = []
    for body in re.findall(r"enum\s*\w*\s*\{(.*?)\}\s*;", text, flags=re.S):
        for item in body.split(","):
            item = item.strip()
            if item:
                nm, val = [x.strip() for x in item.split("=")]
                enums.append((nm, int(val, 0)))
    text = re.sub(r"typedef\s+struct\s+\w*\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    text = re.sub(r"enum\s*\w*\s*\{.*?\}\s*;", " ", text, flags=re.S)
    funcs = []
    for ret, name, args in re.findall(r"([\w\s\*]+?)\b(tfhe_hip_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret = " ".join(ret.split())
        if ret.startswith("typedef"):
            continue
        params = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                m = re.match(r"^(.*?)(\w+)$", a)
                params.append((m.group(1).strip(), m.group(2)))
        funcs.append((name, ret, params))
    return [o[1] for o in opaque], structs, enums, funcs


def rust_type(ctype, opaque_map):
    # tokens: base type words, then per pointer level `*` optionally followed by `const` (which qualifies THAT pointer)
    toks = re.findall(r"\*|\w+", ctype)
    base_words, levels = [], []  # levels[i] = constness of what pointer level i POINTS TO
    pointee_const = False
    i = 0
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            pointee_const = True
        elif toks[i] != "struct":
            base_words.append(toks[i])
        i += 1
    while i < len(toks):
        assert toks[i] == "*", ctype
        levels.append(pointee_const)
        pointee_const = False
        i += 1
        if i < len(toks) and toks[i] == "const":  # `* const`: the pointer itself is const -> what the NEXT level points to
            pointee_const = True
            i += 1
    base = " ".join(base_words)
    if base in PRIM:
        r = PRIM[base]
    elif base in opaque_map:
        r = opaque_map[base]
    else:
        raise ValueError(f"unknown C type {ctype!r}")
    for c in levels:
        r = ("*const " if c else "*mut ") + r
    return r


def generate():
    opaque_all, structs_all, enums_all, funcs_all = [], {}, [], []
    per_header = []
    for h in HEADERS:
        o, s, e, f = parse_header(os.path.join(ROOT, "include", h))
        opaque_all += [x for x in o if x not in opaque_all]
        structs_all.update(s)
        enums_all += e
        per_header.append((h, f))
        funcs_all += f
    opaque_map = {o: camel(o) for o in opaque_all}
    opaque_map.update({s: camel(s) for s in structs_all})
    out = []
    w = out.append
    w("// ffi.rs -- GENERATED by tools/gen_rust_ffi.py from include/tfhe_hip.h, include/tfhe_hip_strings.h and")
    w("// include/tfhe_hip_experimental.h.  Do not edit: change the headers and run the generator again.")
    w("//")
    w("// Where it goes in the reference: `tfhe/src/hip_backend/ffi.rs`, behind a cargo feature `hip` (the same place")
    w("// `tfhe-cuda-backend`'s bindings take in later tfhe-rs versions).  Conventions are those of the reference's own C API")
    w("// (tfhe/src/c_api/utils.rs:3-28, c_api/shortint/server_key/pbs.rs:17-167): every function returns `c_int`, 0 = ok, anything")
    w("// else = failure with the message in `tfhe_hip_last_error()` (thread-local); out-parameters are `*mut *mut T`, set to NULL")
    w("// first; handles are opaque and freed by their `*_destroy`; arrays are caller-owned, 8-byte aligned, contiguous `u64`.")
    w("// NOT COMPILED in this repository (no rustc in the build image): tests/test_rust_ffi.py keeps it in step with the headers")
    w("// and with the symbols `libtfhe_hip.so` exports.")
    w("#![allow(non_camel_case_types, dead_code)]")
    w("use std::os::raw::{c_char, c_int, c_uint, c_void};")
    w("")
    for o in opaque_all:
        w("#[repr(C)]")
        w(f"pub struct {camel(o)} {{")
        w("    _private: [u8; 0],")
        w("}")
    for name, fields in structs_all.items():
        w("")
        w("#[repr(C)]")
        w("#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]")
        w(f"pub struct {camel(name)} {{")
        for ctype, fname in fields:
            w(f"    pub {fname}: {rust_type(ctype, opaque_map)},")
        w("}")
    if enums_all:
        w("")
        for nm, val in enums_all:
            w(f"pub const {nm}: c_int = {val};")
    for h, funcs in per_header:
        w("")
        w(f"// ---- include/{h} ({len(funcs)} functions) " + "-" * max(4, 80 - len(h)))
        w('#[link(name = "tfhe_hip")]')
        w('extern "C" {')
        for name, ret, params in funcs:
            rret = "" if ret == "void" else " -> " + rust_type(ret, opaque_map)
            args = ", ".join(f"{('r#' + a) if a in ('in', 'type', 'fn', 'match', 'ref', 'box', 'move', 'self') else a}: {rust_type(t, opaque_map)}"
                             for t, a in params)
            line = f"    pub fn {name}({args}){rret};"
            if len(line) > 128:
                w(f"    pub fn {name}(")
                for t, a in params:
                    aa = ("r#" + a) if a in ("in", "type", "fn", "match", "ref", "box", "move", "self") else a
                    w(f"        {aa}: {rust_type(t, opaque_map)},")
                w(f"    ){rret};")
            else:
                w(line)
        w("}")
    w("")
    return "\n".join(out), funcs_all


def main():
    text, funcs = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("rust/ffi.rs is stale: run python tools/gen_rust_ffi.py", file=sys.stderr)
            sys.exit(1)
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {len(funcs)} functions")


if __name__ == "__main__":
    main()
