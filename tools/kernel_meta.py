#!/usr/bin/env python3
"""Register / scratch / LDS metadata of every gfx950 kernel in the shipped sources: compiles each translation unit with
`hipcc -S --offload-device-only` (the flags of vuke_amd/csrc/build.sh) and prints the amdhsa metadata of its kernels.

    python tools/kernel_meta.py [--json profiles/archive/r03_kernel_meta.json]

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
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool] + names, capture_output=True, text=True, check=True).stdout.splitlines()
            if len(out) == len(names):
                return [re.sub(r"\(.*", "", o) for o in out]
        except Exception:
            continue
    return names


# issue cost in cycles per wave-instruction per SIMD, per instruction class and per number of resident waves per SIMD:
# profiles/valu_rates.json (tools/ubench_rates.py over tools/ubench_block.hip + tools/ubench_carry.hip: the smallest wall-clock
# figure any member of the class reached at that occupancy) -- what a VALU instruction of each class holds its SIMD for when nothing
# else stalls.  A kernel's mix is priced at ITS occupancy (512 VGPRs / its allocation): the measured
# cycles_per_valu_inst_per_simd (rocprofv3: GRBM_GUI_ACTIVE / SQ_INSTS_VALU) is held against that floor, and bench.py refuses a
# summary whose floor lies above the measurement (round 5 priced three-wave kernels with two-wave rates).
def load_rates():
    path = os.path.join(ROOT, "profiles", "valu_rates.json")
    return {int(w): r for w, r in json.load(open(path))["waves_per_simd"].items()}


def price_for(vgpr, agpr=0, rates=None):
    """(waves per SIMD the allocation allows, {class: cycles} at that occupancy -- the nearest measured one not above)."""
    rates = rates or load_rates()
    waves = max(1, min(8, 512 // max(vgpr + agpr, 1)))
    have = max(w for w in rates if w <= waves)
    return waves, rates[have]


def best_price():
    """{class: cycles}: the smallest issue cost of each class at any occupancy -- what the pipe sustains; the FLOOR of a mix is priced
    with these (tools/ubench_rates.py says why the price at the kernel's own occupancy is not one)."""
    return json.load(open(os.path.join(ROOT, "profiles", "valu_rates.json")))["best_any_occupancy"]


FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_add_nc_u32", "v_min_u32", "v_max_u32", "v_xnor_b32")


def classify(op):
    if op.startswith("v_mad_u64_u32"):
        return "mad64"
    if op.startswith("v_bitop3"):
        return "bitop3"
    if op.startswith("v_mov_b32") or op.startswith("v_accvgpr"):
        return "mov"
    base = op.split("_e32")[0].split("_e64")[0].split("_dpp")[0].split("_sdwa")[0]
    return "fast" if base in FAST else "full"


def valu_mix(txt):
    """Static VALU instruction mix of every kernel body in an assembly listing: {symbol: {class: count}}."""
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^\s+(v_[a-z0-9_]+)", line)
        if cur is not None and m and not m.group(1).startswith(("v_readfirstlane", "v_readlane", "v_writelane")):
            c = classify(m.group(1))
            cur[c] = cur.get(c, 0) + 1
    return out


def unit(src, extra, tag):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "u.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + extra + os.environ.get("VK_EXTRA_FLAGS", "").split() + ["-o", out, src],
                       check=True, capture_output=True)
        txt = open(out).read()
    rows = []
    mix = valu_mix(txt)
    for b in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        row = {"unit": tag, "kernel": name, "vgpr": g("vgpr_count"), "agpr": int(b.split()[0]),
               "sgpr": g("sgpr_count"), "vgpr_spill_count": g("vgpr_spill_count"), "sgpr_spill_count": g("sgpr_spill_count"),
               "scratch_bytes": g("private_segment_fixed_size"), "lds_bytes": g("group_segment_fixed_size")}
        m = mix.get(name)
        if m and sum(m.values()) > 1000:   # the sweep / chain kernels: long straight-line bodies, the static mix is the dynamic one to a few %
            n = sum(m.values())
            row["valu_static_mix"] = {k: round(v / n, 4) for k, v in sorted(m.items())}
            row["valu_static_count"] = n
            waves, price = price_for(row["vgpr"], row["agpr"])
            row["waves_per_simd"] = waves
            best = best_price()
            row["cycles_per_valu_inst_floor"] = round(sum(best[k] * v for k, v in m.items()) / n, 3)
            row["cycles_per_valu_inst_priced_at_occupancy"] = round(sum(price[k] * v for k, v in m.items()) / n, 3)
        rows.append(row)
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
        print(f"{r['unit']:9s} {r['kernel'][:64]:64s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} spills {r['vgpr_spill_count']:3d} scratch {r['scratch_bytes']:5d} B  lds {r['lds_bytes']:6d} B"
              + (f"  floor {r['cycles_per_valu_inst_floor']:.2f} cyc/VALU inst" if "cycles_per_valu_inst_floor" in r else ""))
    if args.json:
        sys.path.insert(0, ROOT)
        import bench
        json.dump({"tool": "tools/kernel_meta.py", "csrc_hash": bench.csrc_hash(), "kernels": rows}, open(args.json, "w"), indent=1)
        print("wrote", args.json)


if __name__ == "__main__":
    main()
