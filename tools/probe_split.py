"""A/B of sweep-kernel variants on the bench geometry (VUKE_HIP_LIB selects the library; written for round 4's split-pass-2
experiment, profiles/archive/r04_split_pass2_experiment.txt): kernel time and wall time per 2^28-key
block of config 2 (Milksad, compressed) and per 2^25-input block of config 3 (sha256, both hash160s, 1 M-slot probe table),
after one untimed block (a variant may size its workspace on first use).  Digests are printed so that two variants can be
compared: they must be equal.

    VUKE_HIP_LIB=vuke_amd/lib/variants/libvuke_hip_split.so python3 tools/probe_split.py [blocks=4] [blocks_per_cu=2]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi
import bench_workloads as wl

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bpc = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # resident workgroups per CU the grid and the workspace are sized for
# PROBE_GEOM="layout,window_bits,batch": another table geometry (default: the bench's nine signed windows, batch 256)
geom = [int(x) for x in os.environ.get("PROBE_GEOM", "1,0,256").split(",")]
eng = vuke_amd.Engine(0, layout=geom[0], window_bits=geom[1], batch=geom[2], blocks_per_cu=bpc)
tag = os.path.basename(os.environ.get("VUKE_HIP_LIB", "shipped"))
rows = [("c2 milksad", ffi.spec(ffi.T_MILKSAD), 1, 28, 1), ("c3 sha256", ffi.spec(ffi.T_SHA256), 3, 25, 3)]
if os.environ.get("PROBE_MORE"):   # config 4 (all four LCG variants, compressed keys: 2^26 inputs = 2^28 keys per block) and a Milksad scan that hashes both key forms
    rows = [("c4 lcg", ffi.spec(ffi.T_LCG), 1, 26, 4), ("milksad both forms", ffi.spec(ffi.T_MILKSAD), 3, 27, 1)]
if os.environ.get("PROBE_MT64"):
    rows = [("mt64", ffi.spec(ffi.T_MT64), 1, 28, 1)]
for name, spec, flags, log2, per in rows:
    if name.startswith("c3"):
        h, k = zip(*wl.c3_decoys())
        eng.targets_set(list(h), list(k))
    n = 1 << log2
    eng.sweep(spec, 0, n - 1, flags=flags)
    ks = wall = 0.0
    keys = 0
    dg = None
    for b in range(1, blocks + 1):
        t = time.perf_counter()
        st, _ = eng.sweep(spec, b * n, (b + 1) * n - 1, flags=flags)
        wall += time.perf_counter() - t
        ks += st["kernel_seconds"]
        keys += st["keys"]
        dg = st["digest"]
    print(f"{tag:32s} geom={geom} bpc={bpc} {name}: {keys / wall / 1e9:.4f} G keys/s wall, {keys / ks / 1e9:.4f} G keys/s kernel, {1e3 * wall / blocks:.2f} ms/block, "
          f"clock {st['shader_clock_ghz']:.3f} GHz, last digest {dg[0]:016x}", flush=True)
eng.close()
