#!/bin/bash
# Counter comparison of sweep-kernel forms on one box (round 6): rocprofv3 --pmc passes over tools/probe_split.py, one block per
# kernel, for a list of "<name>:<library or ->:<blocks_per_cu>[:PROBE_MORE]" cases; tools/pmc_forms.py prints the counters per key.
#   tools/pmc_forms.sh <outdir> <case> [<case> ...]
set -uo pipefail
export TMPDIR=/tmp
out="$1"; shift
mkdir -p "$out"
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT"
for c in "$@"; do
    IFS=: read -r name lib bpc more <<< "$c"
    if [ "$lib" != "-" ]; then export VUKE_HIP_LIB="$lib"; else unset VUKE_HIP_LIB; fi
    if [ -n "${more:-}" ]; then export PROBE_MORE=1; else unset PROBE_MORE; fi
    i=0; mkdir -p "$out/$name"
    for P in "$P1" "$P2" "$P3"; do
        i=$((i+1))
        timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$out/$name/p$i" -o pmc -- python3 tools/probe_split.py 1 "$bpc" > "$out/$name/p$i.log" 2> "$out/$name/p$i.err" || { echo "pass $i of $name failed"; tail -5 "$out/$name/p$i.err"; exit 1; }
    done
    echo "done $name"
done
