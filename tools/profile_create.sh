#!/bin/bash
# rocprofv3 passes over engine creation (the table builder): --kernel-trace --stats for times, separate --pmc passes for counters.
# Usage: tools/profile_create.sh <tag>     -> gpurun_out/prof_<tag>_create/...   (summarise with tools/pmc_create_summary.py)
set -uo pipefail
tag="${1:-r04}"
export TMPDIR=/tmp
root="gpurun_out/prof_${tag}_create"
mkdir -p "$root"
P="python3 tools/probe_create_split.py 1"
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/stats" -o stats -- $P > "$root/stats.out" 2> "$root/stats.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$root/pmc_sq" -o pmc -- $P > "$root/pmc_sq.out" 2> "$root/pmc_sq.log"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$root/pmc_mem" -o pmc -- $P > "$root/pmc_mem.out" 2> "$root/pmc_mem.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$root/pmc_fetch" -o pmc -- $P > "$root/pmc_fetch.out" 2> "$root/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$root/pmc_write" -o pmc -- $P > "$root/pmc_write.out" 2> "$root/pmc_write.log"
du -sh "$root"
