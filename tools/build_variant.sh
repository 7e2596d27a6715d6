#!/bin/bash
# A/B variant of libvuke_hip.so that differs from the shipped build only in some groups of sweep kernels:
#   tools/build_variant.sh <name> "<extra -D flags>" ["<groups>"]   -> vuke_amd/lib/variants/libvuke_hip_<name>.so
# groups: which vk_sweep_tu.hip groups to rebuild with the flags (default "0": Milksad, MT64, raw keys; 1 = sha256 .. direct,
# 2 = lcg, xorshift).  The other translation units are taken from the last full build (vuke_amd/csrc/build/*.o).
set -euo pipefail
name="$1"; flags="${2:-}"; groups="${3:-0}"
root="$(cd "$(dirname "$0")/.." && pwd)"
obj="/tmp/vk_var_$name"; mkdir -p "$obj" "$root/vuke_amd/lib/variants"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
b="$root/vuke_amd/csrc/build"
objs=("$b/vk_engine.o")
pids=()
for g in 0 1 2; do
    if [[ " $groups " == *" $g "* ]]; then
        "$HIPCC" $flags --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -mllvm -pragma-unroll-threshold=200000 \
            -DVK_SWEEP_GROUP=$g -c -o "$obj/vk_sweep_g$g.o" "$root/vuke_amd/csrc/vk_sweep_tu.hip" & pids+=($!)
        objs+=("$obj/vk_sweep_g$g.o")
    else
        objs+=("$b/vk_sweep_g$g.o")
    fi
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$root/vuke_amd/lib/variants/libvuke_hip_$name.so" "${objs[@]}"
echo "built libvuke_hip_$name.so"
# the hooks twin (same sweep objects + the engine compiled with -DVK_TEST_HOOKS), so that tools which look inside an engine
# (probe_ring.py, fuzz_geometries.py) measure the variant and not the shipped kernels
if [ -f "$b/vk_engine_hooks.o" ]; then
    objs[0]="$b/vk_engine_hooks.o"
    "$HIPCC" --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic-functions -o "$root/vuke_amd/lib/variants/libvuke_hip_${name}_hooks.so" "${objs[@]}"
    echo "built libvuke_hip_${name}_hooks.so"
fi
