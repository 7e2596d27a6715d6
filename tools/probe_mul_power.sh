#!/bin/bash
# 256 x 256 -> 512-bit product on v_mad_u64_u32 against v_fma_f64 (tools/ubench_mul.hip): cycles per product, then package power
# and clocks (rocm-smi, read-only) while each variant loops for ~12 s.  usage (GPU box): bash tools/probe_mul_power.sh > out.txt
set -u
tools/ubench_mul
for v in mad fma; do
    echo "=== $v looping"
    tools/ubench_mul $v 12 > /tmp/ubm_$v.txt &
    pid=$!
    sleep 2
    while kill -0 $pid 2>/dev/null; do
        rocm-smi --showpower --showclocks 2>&1 | grep -E "Package Power|sclk" | tr -s '\t ' ' ' | tr '\n' '|'
        echo
        sleep 2
    done
    wait $pid
    cat /tmp/ubm_$v.txt
done
