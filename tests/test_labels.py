"""Label semantics of the boundary, pinned mechanically.

`tests/golden/labels.json` (oracle-made, tests/golden/make_labels.py; the variant names in it were compared with the reference's own
`name()` strings when it was generated) says what `Transform::apply_batch` appends to the input in the label of every slot
(src/transform/lcg.rs:54-66, src/transform/xorshift.rs:43-47 with XorshiftVariant::name() src/xorshift.rs:36-43,
src/transform/sha256_chain.rs:67,88).  Three implementations are held to it:

* the oracle (regenerated here and compared with the committed file),
* the C++ host (vuke_amd/host/transform.cpp, through vkh_label / vkh_parse_transform of libvuke_host.so),
* the Rust binding crate's `label_suffix` (bindings/rust/src/hip_range.rs).  No rustc exists in this pipeline, so the function is
  REPLAYED from its source text: the three name tables and the format strings of the match arms are parsed out of the file, the
  arithmetic of the sha256_chain arm is required to be the text this replay transliterates, and the result is compared slot by slot.
  (Round 3's crate printed "{seed}:64" for an all-variants xorshift hit; nothing could notice.)
"""
import ctypes as C
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "labels.json")))
RS = open(os.path.join(ROOT, "bindings", "rust", "src", "hip_range.rs")).read()
ALL = 0xFFFFFFFF


def test_labels_fixture_is_the_oracles(vko):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_labels", os.path.join(ROOT, "tests", "golden", "make_labels.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.rows() == GOLD["rows"]
    # the reference's spellings, as recorded when the fixture was made in the build container
    assert GOLD["reference_names_checked"]["xorshift"] == ["xorshift64", "xorshift128", "xorshift128+", "xoroshiro128**"]
    kinds = {r["kind"] for r in GOLD["rows"]}
    assert {"lcg", "xorshift", "sha256_chain", "sha256", "milksad", "mt64", "armory"} <= kinds


def rust_fn():
    body = RS[RS.index("pub fn label_suffix("):]
    return body[:body.index("\n}\n") + 3]


def rust_tables(fn):
    t = {}
    for name, items in re.findall(r'const (\w+): \[&str; 4\] = \[(.*?)\];', fn):
        t[name] = re.findall(r'"([^"]*)"', items)
        assert len(t[name]) == 4, name
    return t


def rust_label_suffix(kind_id, variant, endian, depth, sub, ffi):
    """`label_suffix(spec, sub)` of bindings/rust/src/hip_range.rs, replayed from the parsed source."""
    fn = rust_fn()
    tb = rust_tables(fn)
    assert set(tb) == {"LCG", "XS", "CH"}
    assert "let all = spec.variant == VK_VARIANT_ALL;" in fn
    m_lcg = re.search(r'VK_T_LCG if all => format!\("([^"]*)", LCG\[\(sub & 3\) as usize\], if spec\.endian != 0 \{ "([^"]*)" \} else \{ "([^"]*)" \}\),', fn)
    m_xs = re.search(r'VK_T_XORSHIFT if all => format!\("([^"]*)", XS\[\(sub & 3\) as usize\]\),', fn)
    m_ch = re.search(r'VK_T_SHA256_CHAIN => \{\s*let \(nv, d\) = \(if all \{ 4 \} else \{ 1 \}, spec\.chain_depth\.max\(1\)\);\s*let r = sub % \(nv \* d\);\s*'
                     r'let v = if all \{ r / d \} else \{ spec\.variant \};\s*format!\("([^"]*)", CH\[\(v & 3\) as usize\], r % d\)\s*\}', fn)
    assert m_lcg and m_xs and m_ch, "label_suffix changed shape: update this replay together with it"
    assert re.search(r'_ => String::new\(\),', fn)
    fmt = lambda f, *a: f.replace("{}", "%s") % a
    all_ = variant == ALL
    if kind_id == ffi.T_LCG and all_:
        return fmt(m_lcg.group(1), tb["LCG"][sub & 3], m_lcg.group(2) if endian != 0 else m_lcg.group(3))
    if kind_id == ffi.T_XORSHIFT and all_:
        return fmt(m_xs.group(1), tb["XS"][sub & 3])
    if kind_id == ffi.T_SHA256_CHAIN:
        nv, d = (4 if all_ else 1), max(depth, 1)
        r = sub % (nv * d)
        v = r // d if all_ else variant
        return fmt(m_ch.group(1), tb["CH"][v & 3], r % d)
    return ""


def test_rust_crate_label_suffix_equals_the_oracles_labels():
    from vuke_amd import _ffi as ffi
    checked = 0
    for r in GOLD["rows"]:
        for sub, want in enumerate(r["suffix"]):
            assert rust_label_suffix(r["kind_id"], r["variant"], r["endian"], r["chain_depth"], sub, ffi) == want, (r["kind"], r["variant"], r["endian"], sub)
            checked += 1
    assert checked > 250
    # the kind constants the arms are keyed on are the header's (tests/test_abi_layout.py holds the crate's constants to the header)
    assert (ffi.T_LCG, ffi.T_XORSHIFT, ffi.T_SHA256_CHAIN) == (3, 4, 5)


@pytest.fixture(scope="module")
def host():
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "vuke_amd", "host")], check=True)
    L = C.CDLL(os.environ.get("VUKE_HOST_LIB") or os.path.join(ROOT, "vuke_amd", "lib", "libvuke_host.so"))   # VUKE_HOST_LIB: the sanitizer build (tools/run_sanitizers.sh)
    return L


def test_cpp_host_labels_equal_the_oracles_labels(host):
    from vuke_amd import _ffi as ffi
    host.vkh_label.argtypes = [C.POINTER(ffi.TransformSpec), C.c_uint64, C.c_uint32, C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(256)
    for r in GOLD["rows"]:
        sp = ffi.spec(r["kind_id"], r["variant"], r["endian"], r["chain_depth"])
        for seed in (0, 7, 2**64 - 1):
            for sub, want in enumerate(r["suffix"]):
                n = host.vkh_label(C.byref(sp), seed, sub, buf, len(buf))
                assert n >= 0 and buf.value.decode() == f"{seed}{want}", (r["kind"], r["variant"], r["endian"], sub)
