#!/bin/bash
# Builds libvuke_hip.so (HIP kernels + C ABI) for gfx950, in-tree.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -pragma-unroll-threshold: the SHA-512 round loops (2 x 80 rounds in one body) must unroll completely so
# the message schedule stays in registers
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wall -Wno-unused-function -Wno-unused-variable \
    -mllvm -pragma-unroll-threshold=200000 \
    -o "$out/libvuke_hip.so" "$here/vk_engine.hip"
echo "built $out/libvuke_hip.so"
