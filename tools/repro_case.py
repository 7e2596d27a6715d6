#!/usr/bin/env python3
"""Re-run one fuzz_parity case on every engine geometry: tools/repro_case.py kind variant endian depth lo hi flags"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi
from oracle import vko
kind, variant, endian, depth, lo, hi, flags = (int(x) for x in sys.argv[1:8])
vko.build()
gs, os_ = ffi.spec(kind, variant, endian, depth), vko.spec(kind, variant, endian, depth)
ost, _ = vko.sweep(os_, lo, hi, flags=flags, nthreads=16)
for name, kw in (("default", {}), ("signed9 b256", dict(layout=1, batch=256, blocks_per_cu=2)), ("w13 b24", dict(window_bits=13, batch=24, blocks_per_cu=3))):
    e = vuke_amd.Engine(0, **kw)
    st, _ = e.sweep(gs, lo, hi, flags=flags)
    print(os.environ.get("VUKE_HIP_LIB", "default").split("/")[-1], name, (st["inputs"], st["keys"], st["digest"]) == (ost["inputs"], ost["keys"], ost["digest"]), st["keys"], ost["keys"], flush=True)
    e.close()
