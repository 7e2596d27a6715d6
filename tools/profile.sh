#!/bin/bash
# Profiling recipe for the sweep kernel (run on the GPU box through gpurun).
# Usage: tools/profile.sh <tag>     -> writes gpurun_out/prof_<tag>/...
set -uo pipefail
tag="${1:-r01}"
out="gpurun_out/prof_${tag}"
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o stats -- $B --steps 3 --warmup 1 > "$out/bench_stats.json" 2> "$out/stats.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$out/pmc_sq" -o pmc -- $B --steps 1 --warmup 0 > "$out/bench_pmc_sq.json" 2> "$out/pmc_sq.log"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_FLAT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_mem" -o pmc -- $B --steps 1 --warmup 0 > "$out/bench_pmc_mem.json" 2> "$out/pmc_mem.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o pmc -- $B --steps 1 --warmup 0 > "$out/bench_pmc_fetch.json" 2> "$out/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o pmc -- $B --steps 1 --warmup 0 > "$out/bench_pmc_write.json" 2> "$out/pmc_write.log"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$out/pmc_tcc" -o pmc -- $B --steps 1 --warmup 0 > "$out/bench_pmc_tcc.json" 2> "$out/pmc_tcc.log"
find "$out" -name "*.csv" | head -40
du -sh "$out"
