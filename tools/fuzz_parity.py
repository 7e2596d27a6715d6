#!/usr/bin/env python3
"""Randomised parity soak on the GPU box: random transform specs, ranges (around every seed ceiling), flags and engine
geometries; counters + digest of the device sweep against the CPU oracle.  usage: tools/fuzz_parity.py [cases] [seed]"""
import json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi
from oracle import vko

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
vko.build()
threads = min(16, len(os.sched_getaffinity(0)))
engines = [vuke_amd.Engine(0), vuke_amd.Engine(0, layout=1, batch=256, blocks_per_cu=2), vuke_amd.Engine(0, window_bits=13, batch=24, blocks_per_cu=3),
           vuke_amd.Engine(0, layout=1, batch=32, blocks_per_cu=2), vuke_amd.Engine(0, layout=1, window_bits=26, batch=96),
           vuke_amd.Engine(0, layout=1, window_bits=11, batch=8, blocks_per_cu=1)]
edges = [0, 2**31 - 1, 2**31, 2**32 - 1, 2**32, 2**63, 2**64 - 1, 10**9, 10**18]
bad = 0
t0 = time.time()
for c in range(cases):
    kind = rnd.choice([ffi.T_SHA256, ffi.T_MILKSAD, ffi.T_MT64, ffi.T_LCG, ffi.T_XORSHIFT, ffi.T_SHA256_CHAIN, ffi.T_DOUBLE_SHA256, ffi.T_MD5, ffi.T_DIRECT, ffi.T_ARMORY])
    variant = rnd.choice([ffi.VARIANT_ALL, 0, 1, 2, 3]) if kind in (ffi.T_LCG, ffi.T_XORSHIFT, ffi.T_SHA256_CHAIN) else ffi.VARIANT_ALL
    endian = rnd.randrange(2) if kind == ffi.T_LCG else 0
    depth = rnd.choice([1, 2, 7, 10]) if kind == ffi.T_SHA256_CHAIN else 10
    gs, os_ = ffi.spec(kind, variant, endian, depth), vko.spec(kind, variant, endian, depth)
    slots = vko.slots_per_input(os_)
    n = rnd.choice([1, 2, 255, 256, 257, 1000, 65536, rnd.randrange(1, 1 << 17)])
    n = max(1, min(n, (1 << 17) // slots)) if kind != ffi.T_ARMORY else max(1, min(n, 3000))
    centre = rnd.choice(edges + [rnd.getrandbits(rnd.randrange(1, 65))])
    lo = max(0, min(centre - rnd.randrange(0, n + 1), 2**64 - n))
    hi = lo + n - 1
    flags = rnd.choice([1, 2, 3])
    e = rnd.choice(engines)
    st, _ = e.sweep(gs, lo, hi, flags=flags)
    ost, _ = vko.sweep(os_, lo, hi, flags=flags | (vko.F_BATCHED if c % 2 else 0), nthreads=threads)  # both oracle k*G paths
    ok = (st["inputs"], st["keys"], st["digest"]) == (ost["inputs"], ost["keys"], ost["digest"])
    bad += not ok
    if not ok or c % 10 == 9:
        print(json.dumps({"case": c, "ok": ok, "kind": kind, "variant": variant, "endian": endian, "depth": depth, "lo": lo, "hi": hi, "flags": flags,
                          "keys": st["keys"], "elapsed": round(time.time() - t0, 1)}), flush=True)
print(json.dumps({"cases": cases, "failures": bad}))
sys.exit(1 if bad else 0)
