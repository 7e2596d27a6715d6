#!/bin/bash
# Builds libvuke_hip.so (HIP kernels + C ABI) for gfx950, in-tree.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wall -Wno-unused-function \
    -o "$out/libvuke_hip.so" "$here/vk_engine.hip"
echo "built $out/libvuke_hip.so"
