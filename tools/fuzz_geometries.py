#!/usr/bin/env python3
"""Every table geometry the engine accepts (uniform W = 4..26, signed W = 8..29; batch sizes on both sides of the builder's
workspace rule) against the default engine, which the test-suite holds to the oracle: normalised key and both hash160s of edge keys
(digits exactly half / half + 1 / all ones in every window position) and random keys, plus a small Milksad sweep digest.
usage (GPU box): python3 tools/fuzz_geometries.py [keys=1500]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
be = lambda x: x.to_bytes(32, "big")
nkeys = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rnd = random.Random(2026)
ref = vuke_amd.Engine(0)
keys = [be(1), be(2), be(N - 1), be(N - 2), be(N), be(2**256 - 1), bytes(32), be(2**255), be(int("1" * 256, 2) >> 1), be(int("10" * 128, 2)), be(int("01" * 128, 2))]
for sh in range(0, 256, 3):
    keys += [be((1 << sh) % N or 1), be((((1 << 13) - 1) << sh) % N or 1), be((((1 << 29) + 1) << sh) % N or 1)]
keys += [be(rnd.getrandbits(256)) for _ in range(nkeys)]
want = ref.derive_batch(keys)
wst, _ = ref.sweep(ffi.spec(ffi.T_MILKSAD), 5000, 5000 + 2**16 - 1, flags=3)
bad = 0
t0 = time.time()
geoms = [(0, w) for w in range(4, 27)] + [(1, w) for w in range(8, 30)]
for layout, w in geoms:
    for batch in ((3, 200) if w < 27 or layout == 0 else (64,)):
        e = vuke_amd.Engine(0, layout=layout, window_bits=w, batch=batch)
        info = e.info()
        got = e.derive_batch(keys)
        st, _ = e.sweep(ffi.spec(ffi.T_MILKSAD), 5000, 5000 + 2**16 - 1, flags=3)
        ok = got == want and st["digest"] == wst["digest"] and st["keys"] == 2**16
        bad += not ok
        print(f"layout {layout} W {w:2d} batch {batch:3d}: windows {info['windows']:2d} table {info['table_bytes'] / 1e9:8.3f} GB build {1e3 * info['table_build_seconds']:7.2f} ms  "
              f"{'ok' if ok else 'MISMATCH'}  ({time.time() - t0:.0f} s)", flush=True)
        e.close()
print(f"{len(keys)} keys x {sum(2 if (w < 27 or l == 0) else 1 for l, w in geoms)} engines, failures: {bad}")
sys.exit(1 if bad else 0)
