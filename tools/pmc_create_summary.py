#!/usr/bin/env python3
"""Summarise a tools/profile_create.sh run (rocprofv3 over engine creation: the two table-builder kernels) into
profiles/<tag>_create_kernel_stats.csv and profiles/<tag>_create_pmc_summary.json.

usage: tools/pmc_create_summary.py gpurun_out/prof_<tag>_create <tag>
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (csrc_hash only)

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(ROOT, "profiles")
KERNELS = ["vk_table_combine_kernel", "vk_table_seed_kernel"]
ENTRIES = 4 * (2**28 + 1) + 4 * (2**27 + 1) + (2**28 + 1)      # the nine-window signed layout: 120 GB


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


out = {"source": f"profiles/{tag}_create_* (rocprofv3 --kernel-trace --stats and separate --pmc passes of tools/probe_create_split.py 1; tools/profile_create.sh)",
       "csrc_hash": bench.csrc_hash(), "table_entries": ENTRIES, "table_bytes": ENTRIES * 64, "kernels": {}}
stats_csv = find("stats/**/*kernel_stats.csv")
if stats_csv:
    shutil.copy(stats_csv, os.path.join(dst, f"{tag}_create_kernel_stats.csv"))
    for r in csv.DictReader(open(stats_csv)):
        for k in KERNELS:
            if k in r["Name"]:
                out["kernels"].setdefault(k, {})["stats"] = {"calls": int(r["Calls"]), "total_ns": float(r["TotalDurationNs"]), "avg_ns": float(r["AverageNs"])}
counters = {k: collections.defaultdict(float) for k in KERNELS}
meta = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                counters[k][r["Counter_Name"]] += float(r["Counter_Value"])
                meta[k] = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
for k in KERNELS:
    c = dict(counters[k])
    d = out["kernels"].setdefault(k, {})
    d["dispatch"] = meta.get(k)
    d["counters"] = c
    if k == "vk_table_combine_kernel" and "SQ_INSTS_VALU" in c:
        d["valu_insts_per_entry"] = c["SQ_INSTS_VALU"] * 64 / ENTRIES
        if "GRBM_GUI_ACTIVE" in c:
            cycles = c["GRBM_GUI_ACTIVE"] / 8      # summed over the 8 XCDs
            d["gpu_cycles"] = cycles
            d["cycles_per_valu_inst_per_simd"] = cycles * 1024 / c["SQ_INSTS_VALU"]
            if "SQ_ACTIVE_INST_VALU" in c:
                d["valu_busy_pct"] = 100.0 * c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cycles
            if "stats" in d:
                d["effective_clock_ghz"] = cycles / (d["stats"]["total_ns"] * 1e-9) / 1e9
        if "stats" in d:
            sec = d["stats"]["total_ns"] * 1e-9
            d["executed_lane_ops_per_s"] = c["SQ_INSTS_VALU"] * 64 / sec
            d["executed_frac_of_39.3T"] = d["executed_lane_ops_per_s"] / 39.32e12
            d["entries_per_s"] = ENTRIES / sec
            d["algorithmic_bytes"] = ENTRIES * (64 + 64)     # the entry + the parked running product (32 B written, 32 B read)
            d["algorithmic_tb_per_s"] = d["algorithmic_bytes"] / sec / 1e12
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        # FETCH_SIZE / WRITE_SIZE count in KiB; coalesced 16-B-per-lane streams are reported at half their bytes on this part
        # (profiles/fetch_calibration.json: factor 2.00), the builder's reads and writes are such streams
        d["hbm_fetch_bytes_raw"] = c.get("FETCH_SIZE", 0) * 1024
        d["hbm_write_bytes_raw"] = c.get("WRITE_SIZE", 0) * 1024
json.dump(out, open(os.path.join(dst, f"{tag}_create_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
