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
