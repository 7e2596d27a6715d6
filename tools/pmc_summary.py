#!/usr/bin/env python3
"""Turn a tools/profile.sh output directory into profiles/<tag>_* files and profiles/pmc_summary.json.

usage: tools/pmc_summary.py gpurun_out/prof_<tag> <tag> [kernel-substring]
"""
import collections
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
kern = sys.argv[3] if len(sys.argv) > 3 else "vk_sweep_kernel<1u>"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench_stats.json")):
    shutil.copy(os.path.join(src, "bench_stats.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))

counters = collections.defaultdict(float)
launches = collections.defaultdict(set)
meta = {}
for d in sorted(os.listdir(src)):
    f = os.path.join(src, d, "pmc_counter_collection.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            counters[r["Counter_Name"]] += float(r["Counter_Value"])
            launches[r["Counter_Name"]].add(r["Dispatch_Id"])
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
per_launch = {k: v / max(1, len(launches[k])) for k, v in counters.items()}

avg_ns = max_ns = calls = None
for r in csv.DictReader(open(os.path.join(src, "stats", "stats_kernel_stats.csv"))):
    if kern in r["Name"]:
        avg_ns, max_ns, calls = float(r["AverageNs"]), float(r["MaxNs"]), int(r["Calls"])

keys_per_launch = 1 << 28  # the PMC passes run bench.py --steps 1 --warmup 0: one full 2^28-seed launch
out = {
    "source": f"profiles/{tag}_* (rocprofv3 --kernel-trace --stats and separate --pmc passes, tools/profile.sh)",
    "kernel": kern,
    "keys_per_launch": keys_per_launch,
    "kernel_stats": {"calls": calls, "avg_ns_incl_short_warmup_launch": avg_ns, "ns_full_2p28_launch": max_ns},
    "dispatch": meta,
    "counters_per_launch": per_launch,
}
if "SQ_INSTS_VALU" in per_launch:
    out["valu_insts_per_key"] = per_launch["SQ_INSTS_VALU"] / (keys_per_launch / 64)
if "GRBM_GUI_ACTIVE" in per_launch and "SQ_INSTS_VALU" in per_launch:
    cycles = per_launch["GRBM_GUI_ACTIVE"] / 8  # the counter is summed over the 8 XCDs
    out["gpu_cycles_per_launch"] = cycles
    out["effective_clock_ghz"] = cycles / (max_ns * 1e-9) / 1e9
    # issue cost: a VALU wave-instruction holds its SIMD 2 cycles (simple VOP1/VOP2) or 4 (VOP3, carries,
    # multiplier) -- profiles/r01_ubench_valu_rates.txt; 1024 SIMDs on the chip
    out["cycles_per_valu_inst_per_simd"] = cycles * 1024 / per_launch["SQ_INSTS_VALU"]
if "FETCH_SIZE" in per_launch and "WRITE_SIZE" in per_launch:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads 1/2 of the
    # bytes of a wide (16 B/lane) coalesced stream; the table gathers here are 4 x dwordx4 per lane over one 64-byte
    # entry (not a contiguous wave-wide stream), so the raw and the x2 figure are both kept.
    out["hbm_fetch_bytes_raw"] = per_launch["FETCH_SIZE"] * 1024
    out["hbm_write_bytes"] = per_launch["WRITE_SIZE"] * 1024
    out["hbm_bytes_per_launch"] = out["hbm_fetch_bytes_raw"] + out["hbm_write_bytes"]
    out["hbm_bytes_per_launch_fetch_x2"] = 2 * out["hbm_fetch_bytes_raw"] + out["hbm_write_bytes"]
    out["hbm_bytes_per_key_raw"] = out["hbm_bytes_per_launch"] / keys_per_launch
    out["hbm_gbps_raw"] = out["hbm_bytes_per_launch"] / (max_ns * 1e-9) / 1e9
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
