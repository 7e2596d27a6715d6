#!/bin/bash
# A/B variant of libvuke_hip.so that differs from the shipped build only in the group-0 sweep kernels (Milksad, MT64, raw keys):
#   tools/build_variant.sh <name> "<extra -D flags>"   -> vuke_amd/lib/variants/libvuke_hip_<name>.so
# The other translation units are taken from the last full build (vuke_amd/csrc/build/*.o).
set -euo pipefail
name="$1"; flags="${2:-}"
root="$(cd "$(dirname "$0")/.." && pwd)"
obj="/tmp/vk_var_$name"; mkdir -p "$obj" "$root/vuke_amd/lib/variants"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" $flags --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -mllvm -pragma-unroll-threshold=200000 \
    -DVK_SWEEP_GROUP=0 -c -o "$obj/vk_sweep_g0.o" "$root/vuke_amd/csrc/vk_sweep_tu.hip"
b="$root/vuke_amd/csrc/build"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$root/vuke_amd/lib/variants/libvuke_hip_$name.so" "$b/vk_engine.o" "$obj/vk_sweep_g0.o" "$b"/vk_sweep_g[1-2].o
echo "built libvuke_hip_$name.so"
