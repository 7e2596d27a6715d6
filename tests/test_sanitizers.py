"""The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer ("ASAN-enabled build for the
host").  `make asan` builds vuke_amd/host/*.cpp and oracle/*.c with -fsanitize=address,undefined -fno-sanitize-recover=all; here a
bounded part of tools/run_sanitizers.sh runs with the CPU suite (the full run -- 10 000 fuzz cases per parser and every host / oracle /
label test against the sanitizer builds -- is recorded in profiles/r06_sanitizers.txt).  Never on the GPU box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


@pytest.fixture(scope="module")
def asan_build():
    import __graft_entry__ as ge
    ge.build()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "vuke_amd", "host"), "asan"], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    pre = ":".join(subprocess.run(["gcc", "-print-file-name=" + n], capture_output=True, text=True, check=True).stdout.strip() for n in ("libasan.so", "libubsan.so"))
    return dict(ENV, LD_PRELOAD=pre, VKO_LIB=os.path.join(ROOT, "oracle", "build", "asan", "libvko.so"),
                VUKE_HOST_LIB=os.path.join(ROOT, "vuke_amd", "lib", "asan", "libvuke_host.so"), VUKE_HIP_EXE=os.path.join(ROOT, "vuke_amd", "lib", "asan", "vuke-hip"))


def test_byte_level_fuzz_of_the_user_facing_parsers_under_asan_ubsan(asan_build, tmp_path):
    """Targets file -> Matcher::load (reference src/matcher.rs:45-59), wordlist / stdin lines -> LineReader (src/source/stdin.rs:45-60),
    parse_derivation_path (src/bitimage.rs:77-117) and the small grammars beside them: 2 500 mutated inputs each, no sanitizer report, every
    return value in its documented range (tests/fuzz/fuzz_host_parsers.cpp)."""
    r = subprocess.run([os.path.join(ROOT, "vuke_amd", "lib", "asan", "fuzz_host_parsers"), "2500", "606", str(tmp_path)], capture_output=True, text=True, env=ENV, timeout=900)
    assert r.returncode == 0 and "no sanitizer report" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])


def test_host_layer_and_oracle_tests_run_clean_against_the_sanitizer_builds(asan_build):
    """The host-layer, label and (the quick ones of the) oracle tests once more, with libvuke_host.so / vuke-hip / libvko.so replaced by
    their sanitizer builds and python started under LD_PRELOAD=libasan:libubsan."""
    tests = ["tests/test_host_layer.py", "tests/test_labels.py", "tests/test_oracle_hashes.py", "tests/test_oracle_derive.py", "tests/test_oracle_transforms.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", *tests, "-x", "-q", "-p", "no:cacheprovider"], capture_output=True, text=True, env=asan_build, cwd=ROOT, timeout=1500)
    assert r.returncode == 0 and " passed" in r.stdout and "runtime error" not in r.stdout + r.stderr and "AddressSanitizer" not in r.stdout + r.stderr, \
        (r.returncode, r.stdout[-1500:], r.stderr[-3000:])


def test_the_sanitizer_record_is_clean():
    rec = open(os.path.join(ROOT, "profiles", "r06_sanitizers.txt")).read()
    assert rec.count("no sanitizer report") == 2 and " passed" in rec
    assert not any(w in rec for w in ("FAILED", "runtime error", "AddressSanitizer", " failed"))
