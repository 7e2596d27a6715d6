#!/usr/bin/env python3
"""One-off large parity check on the GPU box: the sweep digest and counters of a 2^N-seed Milksad window (both hash160
forms) against the CPU oracle on all host cores.  usage: tools/soak_digest.py [log2 seeds] [first seed]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi
from oracle import vko

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 3_000_000_000
hi = lo + (1 << lg) - 1
vko.build()
e = vuke_amd.Engine(0, layout=1, batch=256)   # blocks_per_cu 0: the form the engine picks (both key forms: the 768-lane workgroups)
out = []
for kind, okind, name in ((ffi.T_MILKSAD, vko.T_MILKSAD, "milksad"), (ffi.T_LCG, vko.T_LCG, "lcg")):
    t = time.perf_counter(); st, _ = e.sweep(ffi.spec(kind), lo, hi, flags=3); tg = time.perf_counter() - t
    t = time.perf_counter(); ost, _ = vko.sweep(vko.spec(okind), lo, hi, flags=3, nthreads=min(16, len(os.sched_getaffinity(0)))); tc = time.perf_counter() - t
    ok = (st["inputs"], st["keys"], st["digest"]) == (ost["inputs"], ost["keys"], ost["digest"])
    out.append({"transform": name, "seeds": [lo, hi], "keys": st["keys"], "digest": ["%016x" % d for d in st["digest"]], "equal_to_oracle": ok,
                "gpu_seconds": tg, "cpu_seconds": tc, "cpu_threads": min(16, len(os.sched_getaffinity(0)))})
    print(json.dumps(out[-1]), flush=True)
sys.exit(0 if all(o["equal_to_oracle"] for o in out) else 1)
