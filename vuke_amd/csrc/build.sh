#!/bin/bash
# Builds libvuke_hip.so (HIP kernels + C ABI) for gfx950, in-tree.  The sweep kernel instantiations are compiled as three
# translation units next to the engine (hipcc jobs in parallel), then linked.  The engine translation unit is compiled a second
# time with -DVK_TEST_HOOKS into libvuke_hip_hooks.so: the same library plus the vk_debug_* hooks of include/vuke_hip_test_hooks.h,
# which tests/ and tools/ load when they look inside an engine; the shipped library exports none of them.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
# VK_OUT / VK_OBJ: build a variant next to the shipped library (same-box A/B through VUKE_HIP_LIB, tools/ab.sh):
#   VK_EXTRA_FLAGS="-DVK_REDUCE_FLAGS=0" VK_OUT=vuke_amd/lib/variants/libvuke_hip_base.so VK_OBJ=/tmp/vk_base bash build.sh
outfile="${VK_OUT:-$here/../lib/libvuke_hip.so}"
out="$(dirname "$outfile")"
obj="${VK_OBJ:-$here/build}"
mkdir -p "$out" "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -pragma-unroll-threshold: the SHA-512 round loops (2 x 80 rounds in one body) must unroll completely so
# the message schedule stays in registers
FLAGS="${VK_EXTRA_FLAGS:-} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -mllvm -pragma-unroll-threshold=200000"
pids=()
"$HIPCC" $FLAGS -c -o "$obj/vk_engine.o" "$here/vk_engine.hip" & pids+=($!)
if [ -z "${VK_NO_HOOKS:-}" ]; then
    "$HIPCC" $FLAGS -DVK_TEST_HOOKS -c -o "$obj/vk_engine_hooks.o" "$here/vk_engine.hip" & pids+=($!)
fi
for g in 0 1 2; do
    "$HIPCC" $FLAGS -DVK_SWEEP_GROUP=$g -c -o "$obj/vk_sweep_g$g.o" "$here/vk_sweep_tu.hip" & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$outfile" "$obj/vk_engine.o" "$obj"/vk_sweep_g[0-2].o
echo "built $outfile"
if [ -z "${VK_NO_HOOKS:-}" ]; then
    hooks="${outfile%.so}_hooks.so"
    "$HIPCC" --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic-functions -o "$hooks" "$obj/vk_engine_hooks.o" "$obj"/vk_sweep_g[0-2].o
    echo "built $hooks"
fi
