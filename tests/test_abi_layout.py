"""The Rust binding crate (bindings/rust) cannot be compiled in this pipeline (no rustc / cargo), so what CAN be
checked is checked mechanically: every `const _: () = assert!(size_of / offset_of ...)` in src/ffi.rs against what a C compiler
says about include/vuke_hip.h, the extern block's function list against the header's declarations (names and argument counts),
the constants against the header's #defines, and the ctypes mirrors of vuke_amd/_ffi.py against the same C numbers."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RS = open(os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")).read()
HDR = open(os.path.join(ROOT, "include", "vuke_hip.h")).read()
NAMES = {"vk_transform_spec": "VkTransformSpec", "vk_hit": "VkHit", "vk_sweep_stats": "VkSweepStats", "vk_engine_opts": "VkEngineOpts",
         "vk_engine_info_t": "VkEngineInfo", "vk_search_target": "VkSearchTarget", "vk_search_result": "VkSearchResult",
         "vk_record_columns": "VkRecordColumns"}


def c_layout(tmp_path):
    exe = tmp_path / "abi_layout"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-o", str(exe),
                    os.path.join(ROOT, "tests", "abi", "abi_layout.c")], check=True, capture_output=True, text=True)
    out = {}
    for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines():
        s, f, n = line.split()
        out[(s, f)] = int(n)
    return out


def test_rust_layout_assertions_equal_the_c_header(tmp_path):
    c = c_layout(tmp_path)
    rust = {}
    for m in re.finditer(r"assert!\(size_of::<(\w+)>\(\) == (\d+)\)", RS):
        rust[(m.group(1), ".")] = int(m.group(2))
    for m in re.finditer(r"assert!\(offset_of!\((\w+), (\w+)\) == (\d+)\)", RS):
        rust[(m.group(1), m.group(2))] = int(m.group(3))
    want = {(NAMES[s], f): n for (s, f), n in c.items() if s in NAMES}
    assert rust == want
    # and every field of every #[repr(C)] struct is asserted: the struct bodies name exactly the asserted fields, in offset order
    for cname, rname in NAMES.items():
        body = re.search(r"pub struct " + rname + r" \{(.*?)\n\}", RS, re.S).group(1)
        fields = re.findall(r"pub (\w+):", body)
        by_off = sorted((n, f) for (s, f), n in c.items() if s == cname and f != ".")
        assert fields == [f for _, f in by_off], rname
    assert int(re.search(r"pub const VK_ABI_VERSION: c_int = (\d+);", RS).group(1)) == c[("VK_ABI_VERSION", ".")]


def test_rust_extern_block_lists_the_headers_functions():
    decl = re.findall(r"^(?:int|uint32_t|size_t|void|const char \*)\s*\**(vk_\w+)\(([^;]*?)\);", HDR, re.M | re.S)
    hdr = {}
    for name, args in decl:
        args = re.sub(r"/\*.*?\*/", "", args, flags=re.S).strip()
        hdr[name] = 0 if args == "void" else len(args.split(","))
    ext = RS[RS.index('extern "C" {'):]
    rs = {}
    for name, args in re.findall(r"pub fn (vk_\w+)\(([^)]*)\)", ext, re.S):
        rs[name] = len([a for a in args.split(",") if a.strip()])
    assert rs == hdr


def _c_to_rust(ctype):
    """Rust spelling of a C parameter / return type of include/vuke_hip.h (the subset the header uses)."""
    t = re.sub(r"/\*.*?\*/", "", ctype, flags=re.S)
    t = re.sub(r"\b[a-z_0-9]+\s*(\[[^\]]*\])?$", lambda m: "*" if m.group(1) else "", t.strip()) if re.search(r"[\w\]]$", t.strip()) and not re.fullmatch(
        r"(const\s+)?(unsigned\s+)?[a-z_0-9]+(\s*\*+)?", t.strip()) else t
    t = " ".join(t.replace("*", " * ").split())
    base = {"int": "c_int", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "uint8_t": "u8", "char": "c_char", "double": "f64",
            "void": "c_void", "vk_engine": "VkEngine"}
    base.update(NAMES)
    toks = t.split()
    const = False
    if toks and toks[0] == "const":
        const, toks = True, toks[1:]
    name, rest = toks[0], toks[1:]
    out = base[name]
    # every '*' adds a pointer level; `const` binds to what is left of it ("const char *const *" = *const *const c_char)
    pending_const = const
    for tok in rest:
        if tok == "*":
            out = ("*const " if pending_const else "*mut ") + out
            pending_const = False
        elif tok == "const":
            pending_const = True
    return out


def test_rust_extern_block_types_equal_the_headers_types():
    """Beyond names and argument counts: every parameter and return TYPE of the extern block is the Rust spelling of the header's C
    type (a `*const` where the header says `const`, `usize` for `size_t`, `u64` for `uint64_t` ...).  The crate cannot be compiled
    here; a wrong width or a dropped `const` would otherwise first show at a maintainer's link time -- or never."""
    decl = re.findall(r"^((?:const\s+)?(?:int|uint32_t|size_t|void|char)\s*\**)\s*(vk_\w+)\(([^;]*?)\);", HDR, re.M | re.S)
    ext = RS[RS.index('extern "C" {'):]
    rs = {}
    for name, args, ret in re.findall(r"pub fn (vk_\w+)\(([^)]*)\)(?:\s*->\s*([^;]+))?;", ext, re.S):
        rs[name] = ([" ".join(a.split(":", 1)[1].split()) for a in args.split(",") if a.strip()], " ".join(ret.split()) if ret else "()")
    checked = 0
    for ret, name, args in decl:
        args = re.sub(r"/\*.*?\*/", "", args, flags=re.S).strip()
        want_args = []
        if args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                m = re.fullmatch(r"(.*?)(\w+)(\s*\[[^\]]*\])?", a)          # type, parameter name, optional array suffix
                ctype = m.group(1).strip() + (" *" if m.group(3) else "")
                want_args.append(_c_to_rust(ctype))
        want_ret = "()" if ret.strip() == "void" else _c_to_rust(ret)
        assert name in rs, name
        assert rs[name][0] == want_args, (name, rs[name][0], want_args)
        assert rs[name][1] == want_ret, (name, rs[name][1], want_ret)
        checked += len(want_args) + 1
    assert checked > 150
    # the translation itself, on the shapes the header uses
    assert [_c_to_rust(t) for t in ("const uint8_t *", "uint8_t *", "const char *const *", "vk_engine **", "size_t *", "const vk_transform_spec *", "uint64_t", "int")] == [
        "*const u8", "*mut u8", "*const *const c_char", "*mut *mut VkEngine", "*mut usize", "*const VkTransformSpec", "u64", "c_int"]


def test_rust_constants_equal_the_headers_defines():
    for name, val in re.findall(r"pub const (VK_[A-Z0-9_]+): \w+ = (-?\w+);", RS):
        if name == "VK_ABI_VERSION":
            continue
        m = re.search(r"#define " + name + r"\s+\(?(-?[0-9xA-Fa-fu]+)\)?", HDR) or re.search(name + r"\s*=\s*(-?\d+)", HDR)
        assert m, name
        assert int(val.replace("_", ""), 0) == int(m.group(1).rstrip("u"), 0), name


def test_ctypes_mirrors_equal_the_c_header(tmp_path):
    from vuke_amd import _ffi as ffi
    c = c_layout(tmp_path)
    for cname, cls in (("vk_transform_spec", ffi.TransformSpec), ("vk_hit", ffi.Hit), ("vk_sweep_stats", ffi.SweepStats), ("vk_engine_opts", ffi.EngineOpts),
                       ("vk_engine_info_t", ffi.EngineInfo), ("vk_search_target", ffi.SearchTarget), ("vk_search_result", ffi.SearchResult),
                       ("vk_record_columns", ffi.RecordColumns)):
        assert ctypes.sizeof(cls) == c[(cname, ".")], cname
        for fname, _ in cls._fields_:
            assert getattr(cls, fname).offset == c[(cname, fname)], (cname, fname)
