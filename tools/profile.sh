#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun).  Separate rocprofv3 passes: --kernel-trace --stats for times,
# --pmc passes for counters (never combined with the trace domains gpurun refuses).  The program follows `--` directly.
# Usage: tools/profile.sh <tag> [configs]      e.g. tools/profile.sh r02a "2 3 4 5m 5e"   -> gpurun_out/prof_<tag>/c<cfg>/...
set -uo pipefail
tag="${1:-r02}"
cfgs="${2:-2}"
export TMPDIR=/tmp
root="gpurun_out/prof_${tag}"
mkdir -p "$root"
for cfg in $cfgs; do
    out="$root/c$cfg"
    mkdir -p "$out"
    case "$cfg" in
        2) steps=3 ;; 3) steps=1 ;; 4) steps=2 ;; 5m) steps=3 ;; 5e) steps=1 ;;
        *) echo "tools/profile.sh: unknown config '$cfg' (2 3 4 5m 5e)"; exit 2 ;;
    esac
    B="python3 bench.py --no-cpu-baseline --no-strong-after --config $cfg"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o stats -- $B --steps $steps --warmup 0 > "$out/bench_stats.json" 2> "$out/stats.log"
    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$out/pmc_sq" -o pmc -- $B --steps 1 --warmup 0 > "$out/bench_pmc_sq.json" 2> "$out/pmc_sq.log"
    rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_FLAT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_mem" -o pmc -- $B --steps 1 --warmup 0 > "$out/bench_pmc_mem.json" 2> "$out/pmc_mem.log"
    # dynamic instruction mix: 32-bit / 64-bit integer VALU instructions (v_mad_u64_u32 and v_lshl_add_u64 count as 64-bit)
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d "$out/pmc_mix" -o pmc -- $B --steps 1 --warmup 0 > "$out/bench_pmc_mix.json" 2> "$out/pmc_mix.log"
    if [ "$cfg" = "2" ]; then
        rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o pmc -- $B --steps 1 --warmup 0 > "$out/bench_pmc_fetch.json" 2> "$out/pmc_fetch.log"
        rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o pmc -- $B --steps 1 --warmup 0 > "$out/bench_pmc_write.json" 2> "$out/pmc_write.log"
        rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$out/pmc_tcc" -o pmc -- $B --steps 1 --warmup 0 > "$out/bench_pmc_tcc.json" 2> "$out/pmc_tcc.log"
    fi
    echo "profiled config $cfg"
done
if [ -x tools/fetch_calib ] && [ "${VK_CALIB:-0}" = "1" ]; then
    mkdir -p "$root/calib"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$root/calib/pmc" -o pmc -- tools/fetch_calib 32 > "$root/calib/fetch_calib.log" 2> "$root/calib/err.log"
    echo "calibration run done"
fi
du -sh "$root"
