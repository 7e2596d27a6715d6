#!/usr/bin/env python3
"""Throughput of the Electrum v1 stretch + child-key kernel (seeds/s, keys/s) on one GPU."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
e = vuke_amd.Engine(0)
rnd = random.Random(5489)
seeds = [("%032x" % rnd.getrandbits(128)).encode() for _ in range(n)]
spec = ffi.spec(ffi.T_ELECTRUM, 0, 0, 20)
e.transform_strings(spec, seeds[:1024])
t = time.perf_counter()
rows = e.transform_strings(spec, seeds)
dt = time.perf_counter() - t
keys = sum(1 for r in rows for k in r if k is not None)
print(f"electrum: {n} 32-hex-char seeds in {dt:.2f} s = {n/dt:,.0f} seeds/s, {keys/dt:,.0f} keys/s (incl. host packing)")
# canonical ops: 200000 SHA-256 compressions x 2168 + 21 x 213000 per seed
print(f"canonical: {n/dt*(200000*2168+21*213000)/1e12:.1f} Tops/s of 39.3")
