"""Where vk_engine_create's wall time goes for the bench geometry (120 GB nine-window table, batch 256, 2 workgroups per CU):
vk_engine_info's split -- HIP runtime start, the two large hipMallocs, the table kernels (hipEvent time and host wall).

    python3 tools/probe_create_split.py [repeats=3] [layout=1] [window_bits=0]      (layout 1 = signed windows: 0 / 29 = nine, 26 = ten, 24 = eleven)

Under rocprofv3 (tools/profile_create.sh) the same run gives the per-kernel times and PMC counters of the table builder."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
layout = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wbits = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kw = dict(layout=1, window_bits=wbits, batch=256, blocks_per_cu=2) if layout == 1 else dict(window_bits=wbits)
for r in range(reps):
    t = time.perf_counter()
    e = vuke_amd.Engine(0, **kw)
    dt = time.perf_counter() - t
    i = e.info()
    print("create %.3f s = runtime %.3f + alloc %.3f + table %.3f (kernels %.4f) + rest %.3f | table %.1f GB, %d windows" % (
        dt, i["runtime_init_seconds"], i["alloc_seconds"], i["table_wall_seconds"], i["table_build_seconds"],
        dt - i["runtime_init_seconds"] - i["alloc_seconds"] - i["table_wall_seconds"], i["table_bytes"] / 1e9, i["windows"]), flush=True)
    t = time.perf_counter()
    e.close()
    print("  destroy %.3f s" % (time.perf_counter() - t), flush=True)
