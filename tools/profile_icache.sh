#!/bin/bash
# Instruction-cache counters of the sweep kernel (run on the GPU box through gpurun): tools/profile_icache.sh <tag> "<configs>"
# -> gpurun_out/prof_<tag>/c<cfg>/pmc_icache.  VUKE_HIP_LIB selects the build.  One --pmc pass, kernel trace only.
set -uo pipefail
tag="${1:-ic}"; cfgs="${2:-2}"
export TMPDIR=/tmp
for cfg in $cfgs; do
    out="gpurun_out/prof_${tag}/c$cfg"; mkdir -p "$out"
    B="python3 bench.py --no-cpu-baseline --no-strong-after --config $cfg"
    rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH --kernel-trace --output-format csv -d "$out/pmc_icache" -o pmc -- $B --steps 1 --warmup 0 > "$out/bench_pmc_icache.json" 2> "$out/pmc_icache.log"
    python3 - "$out" <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc_icache/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vk_sweep_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    req = c.get("SQC_ICACHE_REQ", 0) or 1
    print(k[:40], {n: int(v) for n, v in c.items()}, "miss rate %.4f" % (c.get("SQC_ICACHE_MISSES", 0) / req), "miss+dup rate %.4f" % ((c.get("SQC_ICACHE_MISSES", 0) + c.get("SQC_ICACHE_MISSES_DUPLICATE", 0)) / req))
P
done
