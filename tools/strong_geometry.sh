#!/bin/bash
# Which table geometry gives the shortest END-TO-END 2^32 Milksad sweep on N GPUs?  One GPU plays rank 0 of an N-GPU run
# (bench.py --strong --emulate-gpus N: 16/N blocks), timed from before vk_engine_create to after the counters are reduced.
# Output: one JSON line per (geometry, N) in gpurun_out/<tag>_strong.jsonl
tag="${1:-r02}"
out="gpurun_out/${tag}_strong.jsonl"; : > "$out"
for geo in "--layout 1" "--layout 0 --window-bits 26" "--layout 0 --window-bits 24"; do
  for n in 1 2 4 8 16; do
    timeout -k 10 200 python3 bench.py --strong --emulate-gpus $n $geo --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['strong']
print(json.dumps({'geometry':'$geo','emulated_gpus':$n,'table_gb':round(d['config']['table_gb'],1),'windows':d['config']['windows'],'blocks':s['blocks_per_rank'],'engine_create_s':round(s['engine_create_s'],3),'table_build_s':round(s['table_build_s'],3),'setup_s':round(s['targets_and_fixtures_s'],3),'sweep_s':round(s['sweep_s'],3),'wall_s':round(s['wall_s'],3),'keys_per_s':round(d['value']/1e9,3),'digest_ok':d['digest_ok']}))" | tee -a "$out"
  done
done
