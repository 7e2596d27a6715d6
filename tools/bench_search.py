#!/usr/bin/env python3
"""Analyzer brute force (vk_search_kernel) on one GPU: seeds/s of an exhaustive exact search over the
2^32 Milksad space (target absent: every seed is tried), of a masked LCG search and of a depth-10 sha256_chain search.
Prints one JSON line per search; profiled by tools/profile.sh-style rocprofv3 passes (profiles/archive/r02b_search_*)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi

e = vuke_amd.Engine(0)
absent = bytes(31) + b"\x01"
for name, spec, lo, hi in (("milksad exact", ffi.spec(ffi.T_MILKSAD), 0, 2**32 - 1), ("lcg glibc exact", ffi.spec(ffi.T_LCG, 0, 0), 0, 2**31 - 1),
                           ("sha256_chain exact depth 10", ffi.spec(ffi.T_SHA256_CHAIN), 0, 2**28 - 1)):
    e.search(spec, lo, min(hi, lo + 2**20), key=absent)
    t = time.perf_counter()
    r = e.search(spec, lo, hi, key=absent)
    dt = time.perf_counter() - t
    assert not r["found"]
    print(json.dumps({"search": name, "seeds_checked": r["seeds_checked"], "seconds": dt, "kernel_seconds": r["kernel_seconds"],
                      "seeds_per_s": r["seeds_checked"] / r["kernel_seconds"]}), flush=True)
