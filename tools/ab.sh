#!/bin/bash
# A/B of two builds of libvuke_hip.so on the same GPU box, interleaved: tools/ab.sh <old.so> <new.so> [rounds]
set -euo pipefail
old="$1"; new="$2"; rounds="${3:-3}"
for r in $(seq 1 "$rounds"); do
  for tag in old new; do
    lib="$old"; [ "$tag" = new ] && lib="$new"
    VUKE_HIP_LIB="$lib" python3 bench.py --no-cpu-baseline --no-strong-after --steps 16 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']/1e6,1), 'M keys/s', round(d['ms_per_step'],2), 'ms/step')"
  done
done
