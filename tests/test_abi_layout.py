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


def _header_constants():
    hdr = re.sub(r"/\*.*?\*/", "", HDR, flags=re.S)
    out = {}
    for name, val in re.findall(r"#define (VK_[A-Z0-9_]+)\s+\(?(-?(?:0x[0-9A-Fa-f]+|\d+))u?\)?\s*$", hdr, re.M):
        out[name] = int(val, 0)
    for body in re.findall(r"enum\s*\w*\s*\{(.*?)\}", hdr, re.S):
        nxt = 0
        for item in body.split(","):
            m = re.match(r"\s*(VK_[A-Z0-9_]+)\s*(?:=\s*(-?\w+))?\s*$", item)
            if m:
                nxt = int(m.group(2), 0) if m.group(2) else nxt
                out[m.group(1)] = nxt
                nxt += 1
    return out


def test_every_constant_and_export_of_the_header_has_its_rust_counterpart():
    """the other direction of test_rust_constants_equal_the_headers_defines -- nothing the header defines (every
    VK_E_* error code, VK_T_* transform kind, VK_RC_* record column, flag, layout and mode constant) and nothing it exports may be
    missing from bindings/rust/src/ffi.rs, with the same value; and the crate's `error_kind` names every error code, so that a code
    added to the header (VK_E_NOMEM became reachable from every entry point in round 6) cannot drift past the binding."""
    consts = _header_constants()
    assert {"VK_E_NO_DEVICE", "VK_E_HIP", "VK_E_INVALID_ARG", "VK_E_UNSUPPORTED", "VK_E_NOMEM", "VK_T_SHA256", "VK_T_ARMORY", "VK_RC_COLUMNS", "VK_F_COMPRESSED"} <= set(consts)
    rs = {n: int(v.replace("_", ""), 0) for n, v in re.findall(r"pub const (VK_[A-Z0-9_]+): \w+ = (-?\w+);", RS)}
    for name, val in consts.items():
        if name in ("VK_ABI_VERSION",) or name.startswith("VK_API"):
            continue
        assert name in rs, f"{name} of include/vuke_hip.h has no `pub const` in ffi.rs"
        assert rs[name] == val or rs[name] == val & 0xFFFFFFFF, (name, rs[name], val)
    decl = set(re.findall(r"^(?:const\s+)?(?:int|uint32_t|size_t|void|char)\s*\**\s*(vk_\w+)\(", re.sub(r"/\*.*?\*/", "", HDR, flags=re.S), re.M))
    ext = set(re.findall(r"pub fn (vk_\w+)\(", RS[RS.index('extern "C" {'):]))
    assert decl == ext, (decl ^ ext)
    lib = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    kinds = lib[lib.index("pub fn error_kind"):lib.index("pub fn check")]
    for name in consts:
        if name.startswith("VK_E_"):
            assert f"ffi::{name} =>" in kinds, f"error_kind() in lib.rs does not name {name}"
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in consts:
        if name.startswith("VK_E_"):
            assert name in integ, f"INTEGRATION.md does not list {name}"


def _calls(text):
    return re.findall(r"\b(vk_[a-z_0-9]+)\s*\(", text)


def test_the_c_driver_is_the_rust_sources_call_sequence():
    """tests/abi/hip_range_driver.c executes what bindings/rust/src/hip_range.rs would (tests/test_gpu_parity.py runs it on the GPU box
    against the oracle).  Here the two TEXTS are held to each other: the same vk_* calls in the same order in the scan branch, the
    same retry rule for a hit buffer that was too small, the same sort key, and the same string tables and format strings in
    label_suffix -- so an edit of one side alone fails here, and an edit of both fails on the GPU."""
    rs = open(os.path.join(ROOT, "bindings", "rust", "src", "hip_range.rs")).read()
    c = open(os.path.join(ROOT, "tests", "abi", "hip_range_driver.c")).read()
    c_code = re.sub(r"/\*.*?\*/", "", c, flags=re.S)
    # 1. the call sequence: targets_set, [slots_per_input, sweep], targets_set(clear)
    proc = rs[rs.index("fn process("):]
    scan = proc[:proc.index("} else {")] + proc[proc.index("if matcher.is_some()"):]
    rs_calls = [x for x in _calls(scan)]
    main = c_code[c_code.index("int main("):]
    c_calls = [x for x in _calls(re.sub(r'"(?:\\.|[^"\\])*"', '""', main)) if x not in ("vk_engine_create", "vk_engine_destroy", "vk_last_error")]
    assert rs_calls == c_calls == ["vk_targets_set", "vk_slots_per_input", "vk_sweep", "vk_targets_set"], (rs_calls, c_calls)
    # 2. flags, retry rule, truncate, sort key
    assert "VK_F_COMPRESSED | VK_F_UNCOMPRESSED" in scan and "VK_F_COMPRESSED | VK_F_UNCOMPRESSED" in main
    assert re.search(r"if st\.hits_dropped == 0 \{\s*break;\s*\}", scan) and re.search(r"if \(st\.hits_dropped == 0\) break;", main)
    assert "hits.resize(hits.len() + st.hits_dropped as usize + 1024" in scan and "len = len + (size_t)st.hits_dropped + 1024;" in main
    assert "hits.truncate(st.matches as usize);" in scan and "const size_t matches = (size_t)st.matches;" in main
    assert "hits.sort_by_key(|h| (h.seed, h.sub_index));" in scan and re.search(r"x->seed != y->seed.*x->sub_index < y->sub_index", c_code, re.S)
    assert 'format!("{}{}", h.seed, label_suffix(spec, h.sub_index))' in scan and re.search(r'printf\("%" PRIu64 "%s', main)
    assert "vk_targets_set(e, std::ptr::null(), std::ptr::null(), 0)" in scan and "vk_targets_set(e, NULL, NULL, 0)" in main
    # 3. label_suffix: tables and format strings
    fn_rs = rs[rs.index("pub fn label_suffix"):rs.index("impl Source for")]
    fn_c = c_code[c_code.index("static void label_suffix"):c_code.index("static int by_seed_sub")]
    for tab in ("LCG", "XS", "CH"):
        t_rs = re.search(tab + r": \[&str; 4\] = \[(.*?)\];", fn_rs).group(1)
        t_c = re.search(tab + r"\[4\] = \{(.*?)\};", fn_c).group(1)
        assert re.findall(r'"([^"]*)"', t_rs) == re.findall(r'"([^"]*)"', t_c), tab
    fmts_rs = [f.replace("{}", "%") for f in re.findall(r'format!\("([^"]*)"', fn_rs)]
    fmts_c = [re.sub(r'%s|%" PRIu32 "', "%", f) for f in re.findall(r'snprintf\(out, cap, ("(?:[^"]|" PRIu32 ")*")', fn_c)]
    assert fmts_rs == [f.strip('"') for f in fmts_c] == [":%%", ":%", "[%:%]"], (fmts_rs, fmts_c)
    assert "sub % (nv * d)" in fn_rs and "sub % (nv * d)" in fn_c and "if all { r / d } else { spec.variant }" in fn_rs and "all ? r / d : spec->variant" in fn_c
