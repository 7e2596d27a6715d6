#!/usr/bin/env python3
"""Turn one config directory of a tools/profile.sh run into profiles/<tag>_c<cfg>_* and profiles/pmc_c<cfg>.json (the
summary bench.py reads; it carries the hash of the kernel sources it was taken from and bench.py refuses it otherwise).

usage: tools/pmc_summary.py gpurun_out/prof_<tag> <tag> <cfg>          cfg = 2 | 3 | 4 | 5m | 5e
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (csrc_hash only)

src_root, tag, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
src = os.path.join(src_root, f"c{cfg}")
# the sweep instantiation each config runs with the engine's automatic choice: bench.sweep_kernel_name (held to the engine by
# tests/test_gpu_parity.py through the vk_debug_last_launch hook); PMC_BPC = the --blocks-per-cu the profiled bench ran with
BPC = int(os.environ.get("PMC_BPC", "0"))
KERNELS = {"2": [bench.sweep_kernel_name(1, 1, BPC)], "3": [bench.sweep_kernel_name(0, 3, BPC)], "4": [bench.sweep_kernel_name(3, 1, BPC)],
           "5m": ["vk_ring_kernel<vk_pbkdf2_body>", "vk_mb_prepare_kernel", "vk_mb_middle_kernel", "vk_mb_final_kernel", "vk_sweep_kernel<100u"],
           "5e": ["vk_ring_kernel<vk_electrum_body>", "vk_electrum_prepare_kernel", "vk_electrum_final_kernel", "vk_sweep_kernel<100u"]}[cfg]
dst = os.path.join(ROOT, "profiles")


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


stats_csv = find("stats/**/*kernel_stats.csv")
if stats_csv:
    shutil.copy(stats_csv, os.path.join(dst, f"{tag}_c{cfg}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench_stats.json")):
    shutil.copy(os.path.join(src, "bench_stats.json"), os.path.join(dst, f"{tag}_c{cfg}_bench_under_rocprof.json"))


def bench_line(name):
    p = os.path.join(src, name)
    if not os.path.exists(p):
        return None
    for line in open(p):
        if line.startswith("{"):
            return json.loads(line)
    return None


counters = collections.defaultdict(float)
meta = {}
per_kernel_valu = collections.defaultdict(float)
mix = collections.defaultdict(float)   # the pmc_mix pass counts SQ_INSTS_VALU again: kept apart so that no counter is summed twice
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    is_mix = os.sep + "pmc_mix" + os.sep in f
    for r in csv.DictReader(open(f)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                if is_mix:
                    if k == KERNELS[0]:
                        mix[r["Counter_Name"]] += float(r["Counter_Value"])
                    continue
                counters[r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "SQ_INSTS_VALU":
                    per_kernel_valu[k] += float(r["Counter_Value"])
                if k == KERNELS[0]:
                    meta = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}

kstats = {}
if stats_csv:
    for r in csv.DictReader(open(stats_csv)):
        for k in KERNELS:
            if k in r["Name"]:
                kstats[k] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "max_ns": float(r["MaxNs"]), "total_ns": float(r["TotalDurationNs"])}

# every --pmc pass ran `bench.py --steps 1 --warmup 0`: the counters of one pass cover exactly the keys of that line
sq = bench_line("bench_pmc_sq.json")
keys = sq["roofline"]["keys_per_launch"] * 1 if sq else None
keys_total = sq["keys_total"] if sq and "keys_total" in sq else None
units_total = sq.get("units_total") if sq else None
try:
    git = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    git = None
out = {
    "source": f"profiles/{tag}_c{cfg}_* (rocprofv3 --kernel-trace --stats and separate --pmc passes of `bench.py --config {cfg}`, tools/profile.sh)",
    "csrc_hash": bench.csrc_hash(), "git": git, "config": cfg, "kernels": KERNELS,
    "keys_in_pmc_pass": keys_total, "units_in_pmc_pass": units_total,
    "kernel_stats": kstats, "dispatch": meta, "counters_per_pass": dict(counters),
}
if "SQ_INSTS_VALU" in counters and keys_total:
    out["valu_insts_per_key"] = counters["SQ_INSTS_VALU"] / (keys_total / 64)
    out["valu_insts_per_key_by_kernel"] = {k: v / (keys_total / 64) for k, v in per_kernel_valu.items()}
    if units_total and units_total != keys_total:
        out["valu_insts_per_input"] = counters["SQ_INSTS_VALU"] / (units_total / 64)
if "GRBM_GUI_ACTIVE" in counters and "SQ_INSTS_VALU" in counters and kstats:
    mem = bench_line("bench_pmc_mem.json")
    cycles = counters["GRBM_GUI_ACTIVE"] / 8  # the counter is summed over the 8 XCDs
    kernel_s = (mem["roofline"]["kernel_ms_per_launch"] * 1e-3 * max(1, round(mem["keys_total"] / mem["roofline"]["keys_per_launch"]))) if mem and "keys_total" in mem else None
    out["gpu_cycles_per_pass"] = cycles
    if kernel_s:
        out["effective_clock_ghz"] = cycles / kernel_s / 1e9
    out["cycles_per_valu_inst_per_simd"] = cycles * 1024 / counters["SQ_INSTS_VALU"]
    # rocprof's derived metric VALUBusy = 100 * SQ_ACTIVE_INST_VALU * 4 / SIMD_NUM / GRBM_GUI_ACTIVE (per XCD): the share of
    # cycles in which a SIMD's VALU is executing, at 4 cycles per wave-instruction
    if "SQ_ACTIVE_INST_VALU" in counters:
        out["valu_busy_pct"] = 100.0 * counters["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cycles
if mix.get("SQ_INSTS_VALU"):
    # dynamic mix of the dominant kernel: share of 64-bit integer VALU instructions (the multiplier v_mad_u64_u32, and
    # v_lshl_add_u64 in the SHA-512 kernels); the floor prices that share at the multiplier's issue cost and the rest with the
    # static mix of the code object (tools/kernel_meta.py: cycles per wave-instruction per SIMD with two waves resident,
    # profiles/archive/r03_ubench_valu_rates.txt)
    out["valu_mix"] = {"int64_share": mix.get("SQ_INSTS_VALU_INT64", 0.0) / mix["SQ_INSTS_VALU"], "int32_share": mix.get("SQ_INSTS_VALU_INT32", 0.0) / mix["SQ_INSTS_VALU"]}
    km_path = os.path.join(dst, "kernel_meta.json")
    if os.path.exists(km_path):
        km = json.load(open(km_path))
        row = next((r for r in km["kernels"] if KERNELS[0] in r["kernel"] and "valu_static_mix" in r), None)
        if row and km.get("csrc_hash") == bench.csrc_hash():
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import kernel_meta as kmeta
            waves, price = kmeta.price_for(row["vgpr"], row["agpr"])
            best = kmeta.best_price()
            sm = dict(row["valu_static_mix"])
            sm.pop("mad64", None)
            rest = sum(sm.values())
            share = out["valu_mix"]["int64_share"]

            def priced(pr):
                rest_price = sum(pr[k] * v for k, v in sm.items()) / rest if rest else pr["full"]
                return share * pr["mad64"] + (1.0 - share) * rest_price
            out["waves_per_simd"] = waves
            out["cycles_per_valu_inst_floor"] = priced(best)
            out["cycles_per_valu_inst_priced_at_occupancy"] = priced(price)
            out["cycles_per_valu_inst_floor_note"] = ("dynamic 64-bit share (SQ_INSTS_VALU_INT64 / SQ_INSTS_VALU) + the rest at the static mix of the code object "
                                                      "(profiles/kernel_meta.json), every class at the best issue rate it reaches at ANY occupancy (profiles/valu_rates.json: "
                                                      "tools/ubench_block.hip + tools/ubench_carry.hip) = what the pipes sustain for this mix; `priced_at_occupancy` = the same mix "
                                                      f"at the single-class rates of the kernel's own {waves} waves per SIMD (not a floor: a real mix overlaps its classes across waves)")
            if "cycles_per_valu_inst_per_simd" in out:
                out["issue_efficiency"] = out["cycles_per_valu_inst_floor"] / out["cycles_per_valu_inst_per_simd"]
                w = counters
                if "SQ_WAVE_CYCLES" in w:
                    out["wave_cycle_split"] = {k: w[k] / w["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in w}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters and keys_total:
    fetch, write = counters["FETCH_SIZE"] * 1024, counters["WRITE_SIZE"] * 1024
    out["hbm_fetch_bytes_raw"] = fetch
    out["hbm_write_bytes"] = write
    out["hbm_bytes_per_launch"] = fetch + write
    out["hbm_bytes_per_key_raw"] = (fetch + write) / keys_total
    cal_path = os.path.join(dst, "fetch_calibration.json")
    if os.path.exists(cal_path):
        cal = json.load(open(cal_path))["patterns"]
        # FETCH_SIZE tallies requests at 64 B: the coalesced workspace reads (160 B/key, 128-B requests) show up at 1/s of their
        # bytes, the random table / probe reads at 1/g.  With the workspace stream taken at its algorithmic volume:
        s, g = cal["calib_stream"]["factor"], cal["calib_gather_lds"]["factor"]
        stream = 160.0 * keys_total
        gathers = g * max(0.0, fetch - stream / s)
        algo_gather = (int(sq["config"]["windows"]) * 64 + 32) * keys_total
        out["fetch_calibration"] = {"stream_factor": s, "gather_factor": g, "assumed_stream_read_bytes": stream,
                                    "gather_read_bytes_calibrated": gathers, "algorithmic_gather_bytes": algo_gather,
                                    "gather_overfetch_ratio": gathers / algo_gather}
        out["hbm_bytes_per_launch_calibrated"] = gathers + stream + write
        out["hbm_bytes_per_key_calibrated"] = out["hbm_bytes_per_launch_calibrated"] / keys_total
json.dump(out, open(os.path.join(dst, f"{tag}_c{cfg}_pmc_summary.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, f"pmc_c{cfg}.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
