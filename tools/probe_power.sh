#!/bin/bash
# Samples power, clocks and temperature (rocm-smi, read-only) while bench.py sweeps: why the sweep kernels hold ~2.1-2.2 GHz and the
# SHA-2 chain kernels ~2.35 GHz of the 2.4 GHz the roofline's peak is priced at.  usage (GPU box): bash tools/probe_power.sh > out.txt
set -u
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -v "^$" | head -40
for cfg in 2 5e; do
    echo "=== bench.py --config $cfg running"
    python3 bench.py --no-cpu-baseline --config $cfg --steps $([ "$cfg" = 2 ] && echo 80 || echo 12) --warmup 1 > /tmp/bench_$cfg.json 2> /tmp/bench_$cfg.err &
    pid=$!
    while kill -0 $pid 2>/dev/null; do
        rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Package Power|sclk|Temperature \(Sensor (junction|memory)" | tr -s '\t ' ' ' | tr '\n' '|'
        echo
        sleep 1
    done
    wait $pid
    python3 -c "
import json,sys
d=json.loads(open('/tmp/bench_$cfg.json').read().strip().splitlines()[-1])
print('bench value', d['value'], d['unit'], 'shader_clock_ghz', d['roofline'].get('shader_clock_ghz'))"
done
