"""CPU-side checks of the drop-in boundary: the library loads, exports exactly what
include/vuke_hip.h declares, and refuses to run without a GPU (no CPU fallback)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "vuke_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(vk_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported(built):
    from vuke_amd import _ffi
    L = _ffi.load()
    names = declared_functions()
    assert "vk_sweep" in names and "vk_engine_create" in names and len(names) >= 12
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/vuke_hip.h but not exported"
    assert sorted(_ffi.EXPORTS) == names
    assert L.vk_abi_version() == 5


def declared_hooks():
    hdr = open(os.path.join(ROOT, "include", "vuke_hip_test_hooks.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(vk_[a-z0-9_]+)\s*\(", hdr)))


def test_shipped_library_exports_no_test_hooks(built):
    """The vk_debug_* hooks live in include/vuke_hip_test_hooks.h and in libvuke_hip_hooks.so (the same sources with
    -DVK_TEST_HOOKS) only: the shipped library exports none of them, the test build all of them plus the whole boundary."""
    from vuke_amd import _ffi
    hooks = declared_hooks()
    assert hooks == sorted(_ffi.HOOK_EXPORTS) and len(hooks) == 8 and all(h.startswith("vk_debug_") for h in hooks)
    L, H = _ffi.load(), _ffi.load(hooks=True)
    for h in hooks:
        assert not hasattr(L, h), f"{h} exported by the shipped library"
        assert hasattr(H, h), f"{h} missing from libvuke_hip_hooks.so"
    for n in declared_functions():
        assert hasattr(H, n)
    assert not any("vk_debug" in n for n in declared_functions())
    # nothing in the product reads launch-policy knobs from the environment any more
    eng_src = open(os.path.join(ROOT, "vuke_amd", "csrc", "vk_engine.hip")).read()
    assert "getenv" not in eng_src


def test_hooks_library_is_resolved_on_its_own(built, tmp_path):
    """with VUKE_HIP_LIB naming an A/B variant that has no hooks twin, Engine(hooks=True) must fall back to the in-tree
    hooks build instead of looking for variants/libvuke_hip_<name>_hooks.so; a twin beside the variant, or VUKE_HIP_HOOKS_LIB, wins."""
    import shutil
    import subprocess
    import sys
    in_tree = os.path.join(ROOT, "vuke_amd", "lib", "libvuke_hip_hooks.so")
    variant = tmp_path / "libvuke_hip_x.so"
    shutil.copy(os.path.join(ROOT, "vuke_amd", "lib", "libvuke_hip.so"), variant)

    def resolved(env):
        e = dict(os.environ, **env)
        code = "from vuke_amd import _ffi; _ffi.load(hooks=True); print(_ffi.LIB_PATH); print(_ffi.HOOKS_LIB_PATH)"
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, check=True, capture_output=True, text=True).stdout.split()
        return out

    assert resolved({"VUKE_HIP_LIB": str(variant)}) == [str(variant), in_tree]
    twin = tmp_path / "libvuke_hip_x_hooks.so"
    shutil.copy(in_tree, twin)
    assert resolved({"VUKE_HIP_LIB": str(variant)}) == [str(variant), str(twin)]
    other = tmp_path / "elsewhere_hooks.so"
    shutil.copy(in_tree, other)
    assert resolved({"VUKE_HIP_LIB": str(variant), "VUKE_HIP_HOOKS_LIB": str(other)}) == [str(variant), str(other)]


def test_struct_layouts_match_header():
    import ctypes as C
    from vuke_amd import _ffi
    assert C.sizeof(_ffi.Hit) == 72
    assert C.sizeof(_ffi.SweepStats) == 88
    assert C.sizeof(_ffi.TransformSpec) == 16
    assert C.sizeof(_ffi.EngineOpts) == 32


def test_slots_per_input_without_gpu(built):
    import ctypes as C
    from vuke_amd import _ffi
    L = _ffi.load()
    f = lambda *a: L.vk_slots_per_input(C.byref(_ffi.spec(*a)))
    assert f(_ffi.T_SHA256) == 3 and f(_ffi.T_MILKSAD) == 1 and f(_ffi.T_MT64) == 1
    assert f(_ffi.T_LCG) == 4 and f(_ffi.T_LCG, 2) == 1 and f(_ffi.T_LCG, 9) == 0
    assert f(_ffi.T_XORSHIFT) == 4 and f(_ffi.T_SHA256_CHAIN) == 80 and f(_ffi.T_SHA256_CHAIN, 1, 0, 7) == 14
    assert f(42) == 0


def test_no_cpu_fallback(built):
    """Without a GPU the product path must fail loudly, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import vuke_amd
    with pytest.raises(vuke_amd.VkError) as ei:
        vuke_amd.Engine(0)
    assert ei.value.code == vuke_amd._ffi.VK_E_NO_DEVICE


def test_product_does_not_touch_oracle():
    """Nothing under vuke_amd/ or include/ may reference the oracle."""
    bad = []
    for base in ("vuke_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".cuh", ".hip", ".cpp", ".hpp", ".sh")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"\bvko_|libvko|import oracle|from oracle|oracle/vko", txt):
                        bad.append(os.path.join(dp, f))
    assert bad == []


def test_coalescer_lint_record_covers_the_kernel_sources_in_the_tree():
    """tools/lint_coalescer.py (the mechanical check for the LLVM register-coalescer defect behind round 1's short-fold
    failure, profiles/archive/r02_short_fold_root_cause.txt) was run on exactly these kernel sources and found nothing."""
    import json
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "lint_coalescer.json")))
    assert rec["dropped_outputs"] == 0 and rec.get("not_comparable", 0) == 0 and rec["asm_statements"] > 8000
    assert rec["csrc_hash"] == bench.csrc_hash(), "kernel sources changed: re-run tools/lint_coalescer.py (about 4 minutes)"


def test_kernel_source_hash_ignores_comments_but_not_code(tmp_path, monkeypatch):
    """bench.csrc_hash keys the committed PMC profiles / lint record on the kernel CODE: rewording a comment must not void them, a
    changed token must (string literals count as code)."""
    import bench
    d = tmp_path / "vuke_amd" / "csrc"
    d.mkdir(parents=True)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (d / "a.cuh").write_text('// one\nint f(int x) { return x + 1; } /* two */\nconst char *s = "// not a comment";\n')
    h0 = bench.csrc_hash()
    (d / "a.cuh").write_text('// a different remark\n\nint f(int x)   { return x + 1; }   /* gone\n over lines */\nconst char *s = "// not a comment";\n')
    assert bench.csrc_hash() == h0
    (d / "a.cuh").write_text('// one\nint f(int x) { return x + 2; } /* two */\nconst char *s = "// not a comment";\n')
    assert bench.csrc_hash() != h0
    (d / "a.cuh").write_text('// one\nint f(int x) { return x + 1; } /* two */\nconst char *s = "// not a comment!";\n')
    assert bench.csrc_hash() != h0


def test_profile_files_cited_in_the_documents_exist():
    """DESIGN / README / INTEGRATION cite rocprof summaries and probe records under profiles/ by name; a re-taken profile renames
    them (the tag carries the round and a letter), and a stale citation would point the reader at nothing."""
    import glob
    import re
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("tools", "README.md"), os.path.join("docs", "DESIGN_rounds_1-5.md")):
        text = open(os.path.join(ROOT, doc)).read()
        for cited in set(re.findall(r"profiles/[A-Za-z0-9_.*{},/\-]+", text)):   # sub-directories too: profiles/archive/ holds rounds 1 - 4
            cited = cited.rstrip(".,)")
            pats = [cited]
            while any("{" in p for p in pats):   # every brace group, not only the first
                nxt = []
                for p in pats:
                    m = re.match(r"(.*?)\{([^}]*)\}(.*)", p)
                    nxt += [m.group(1) + x + m.group(3) for x in m.group(2).split(",")] if m else [p.replace("{", "")]
                pats = nxt
            for p in pats:
                assert glob.glob(os.path.join(ROOT, p)) or glob.glob(os.path.join(ROOT, p + "*")), (doc, p)


def test_tests_cited_in_the_documents_exist():
    import glob
    import re
    names = set()
    for f in glob.glob(os.path.join(ROOT, "tests", "*.py")):
        names |= set(re.findall(r"^def (test_\w+)", open(f).read(), re.M))
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for cited in set(re.findall(r"::(test_[A-Za-z0-9_]+\*?)", text)):
            assert any(n.startswith(cited.rstrip("*")) for n in names), (doc, cited)
        for f in set(re.findall(r"tests/(test_\w+\.py)", text)):
            assert os.path.exists(os.path.join(ROOT, "tests", f)), (doc, f)


def test_an_allocation_failure_inside_the_library_is_a_negative_code_not_an_exception(built):
    """include/vuke_hip.h promises that nothing throws across the boundary.  Every extern "C" function of
    vk_engine.hip is a function-try-block (VK_API_CATCH): std::bad_alloc -> VK_E_NOMEM with vk_last_error set.  Here, without a GPU:
    vk_engine_create allocates the engine (nothrow new), reserves its error string and starts the HIP runtime; with the hooks
    build's allocation-failure injection armed for the k-th allocation of this thread the call must come back with VK_E_NOMEM (the
    failure hit one of the library's own allocations) or with what it returns unarmed (the k-th allocation does not exist) -- in a
    child process, so that a std::terminate would be seen as a dead child and not take the test runner with it."""
    code = r'''
import ctypes as C, sys
from vuke_amd import _ffi
H = _ffi.load(hooks=True)
e = C.c_void_p()
base = H.vk_engine_create(0, C.byref(e))          # unarmed: VK_E_NO_DEVICE here, VK_OK on a GPU box
if base == 0:
    H.vk_engine_destroy(e)
seen_nomem = 0
for k in range(0, 6):
    H.vk_debug_fail_alloc(k)
    e = C.c_void_p()
    rc = H.vk_engine_create(0, C.byref(e))
    made = H.vk_debug_fail_alloc(-1)
    msg = (H.vk_last_error(None) or b"").decode()
    assert rc in (base, _ffi.VK_E_NOMEM), (k, rc, msg)
    if rc == _ffi.VK_E_NOMEM:
        seen_nomem += 1
        assert not e.value and "memory" in msg, (k, msg)
    elif rc == 0:
        H.vk_engine_destroy(e)
    print(k, rc, made, msg)
assert seen_nomem >= 2, seen_nomem                   # k = 0 (the engine itself, nothrow new) and k = 1 (a throwing allocation)
print("ok")
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


def test_every_extern_c_function_of_the_engine_is_guarded():
    """The guard is structural: every extern "C" definition in vk_engine.hip that has a body of more than one line either is a
    function-try-block closed by VK_API_CATCH / catch (...), or is one of the few that cannot throw (no allocation, no lock)."""
    src = open(os.path.join(ROOT, "vuke_amd", "csrc", "vk_engine.hip")).read()
    cannot_throw = {"vk_abi_version", "vk_device_count", "vk_slots_per_input", "vk_slots_per_string", "vk_last_error", "vk_engine_destroy",
                    "vk_engine_create", "vk_engine_info", "vk_record_column_stride", "vk_debug_fail_alloc"}
    names = re.findall(r'^extern "C" [\w \*]+?\b(vk_\w+)\(', src, flags=re.M)
    assert len(names) >= 37
    for n in names:
        m = re.search(r'^extern "C" [^\n]*\b' + n + r'\([^{;]*?\)\s*(try)?\s*\{', src, flags=re.M | re.S)
        assert m, n
        if n in cannot_throw:
            continue
        assert m.group(1) == "try", f"{n} is not a function-try-block"
    assert src.count("\nVK_API_CATCH(") >= 26
