#!/usr/bin/env python3
"""Register / scratch / LDS metadata of every gfx950 kernel in the shipped sources: compiles each translation unit with
`hipcc -S --offload-device-only` (the flags of vuke_amd/csrc/build.sh) and prints the amdhsa metadata of its kernels.

    python tools/kernel_meta.py [--json profiles/r03_kernel_meta.json]

(The same numbers `llvm-readelf --notes` shows for the code objects inside libvuke_hip.so; reading them from the assembly
avoids unbundling the fat binary.)"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vuke_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-Wno-unused-variable", "-mllvm", "-pragma-unroll-threshold=200000",
         "-S", "--offload-device-only"]


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True, check=True).stdout.splitlines()
        return [re.sub(r"\(.*", "", o) for o in out]
    except Exception:
        return names


def unit(src, extra, tag):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "u.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + extra + os.environ.get("VK_EXTRA_FLAGS", "").split() + ["-o", out, src],
                       check=True, capture_output=True)
        txt = open(out).read()
    rows = []
    for b in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))
        rows.append({"unit": tag, "kernel": re.search(r"\.name:\s+(\S+)", b).group(1), "vgpr": g("vgpr_count"), "agpr": int(b.split()[0]),
                     "sgpr": g("sgpr_count"), "vgpr_spill_count": g("vgpr_spill_count"), "sgpr_spill_count": g("sgpr_spill_count"),
                     "scratch_bytes": g("private_segment_fixed_size"), "lds_bytes": g("group_segment_fixed_size")})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    args = ap.parse_args()
    jobs = [(os.path.join(CSRC, "vk_sweep_tu.hip"), [f"-DVK_SWEEP_GROUP={g}"], f"sweep_g{g}") for g in range(3)]
    jobs.append((os.path.join(CSRC, "vk_engine.hip"), [], "engine"))
    with ThreadPoolExecutor(max_workers=4) as ex:
        rows = [r for rs in ex.map(lambda j: unit(*j), jobs) for r in rs]
    for r, d in zip(rows, demangle([r["kernel"] for r in rows])):
        r["kernel"] = d
    for r in rows:
        print(f"{r['unit']:9s} {r['kernel'][:64]:64s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} spills {r['vgpr_spill_count']:3d} scratch {r['scratch_bytes']:5d} B  lds {r['lds_bytes']:6d} B")
    if args.json:
        sys.path.insert(0, ROOT)
        import bench
        json.dump({"tool": "tools/kernel_meta.py", "csrc_hash": bench.csrc_hash(), "kernels": rows}, open(args.json, "w"), indent=1)
        print("wrote", args.json)


if __name__ == "__main__":
    main()
