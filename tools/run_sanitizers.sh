#!/bin/bash
# The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer (-fno-sanitize-recover=all), never on the GPU box:
#   * vuke_amd/host/*.cpp (libvuke_host.so, vuke-hip) and oracle/*.c (libvko.so) built with `make asan`,
#   * tests/test_host_layer.py, tests/test_oracle_*.py, tests/test_labels.py run against those builds (python under LD_PRELOAD=libasan),
#   * tests/fuzz/fuzz_host_parsers: N byte-level cases per parser (targets file -> Matcher::load, LineReader, parse_derivation_path, ...).
# usage: tools/run_sanitizers.sh [cases=10000] [record=profiles/r06_sanitizers.txt]
set -uo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
cases="${1:-10000}"; rec="${2:-$root/profiles/r06_sanitizers.txt}"
cd "$root"
make -s -C vuke_amd/host asan && make -s -C oracle asan || exit 1
asan="$(gcc -print-file-name=libasan.so)"; ubsan="$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1" UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
tmp="$(mktemp -d)"; trap 'rm -rf "$tmp"' EXIT
{
    echo "tools/run_sanitizers.sh $cases -- $(date -u +%Y-%m-%dT%H:%MZ), $(gcc --version | head -1), flags: -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all"
    echo
    echo "== byte-level fuzz of the user-facing parsers (tests/fuzz/fuzz_host_parsers.cpp, sanitizer build of libvuke_host.so)"
    for seed in 1 2; do
        vuke_amd/lib/asan/fuzz_host_parsers $((cases / 2)) $seed "$tmp" 2>&1 | tail -5 || echo "FAILED (seed $seed)"
    done
    echo
    echo "== pytest against the sanitizer builds (python preloaded with libasan + libubsan)"
    LD_PRELOAD="$asan:$ubsan" VKO_LIB="$root/oracle/build/asan/libvko.so" VUKE_HOST_LIB="$root/vuke_amd/lib/asan/libvuke_host.so" \
        VUKE_HIP_EXE="$root/vuke_amd/lib/asan/vuke-hip" \
        python3 -m pytest tests/test_host_layer.py tests/test_oracle_transforms.py tests/test_oracle_derive.py tests/test_oracle_hashes.py tests/test_oracle_electrum.py \
        tests/test_oracle_multibit.py tests/test_oracle_bitimage.py tests/test_oracle_search.py tests/test_oracle_openssl.py tests/test_labels.py -x -q -p no:cacheprovider 2>&1 | tail -15
} | tee "$rec"
grep -q "FAILED\|ERROR\|runtime error\|AddressSanitizer" "$rec" && exit 1
exit 0
