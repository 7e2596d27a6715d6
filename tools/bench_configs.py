#!/usr/bin/env python3
"""One-GPU measurements of the other BASELINE.json configs (parity-test cases, not bench lines):
config 3 (scan --transform=sha256 vs 1 M targets), config 4 slice (all LCG variants), plus the
per-transform sweep rates.  Prints one JSON line per measurement."""
import json, os, struct, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi

layout = int(sys.argv[1]) if len(sys.argv) > 1 else 1
e = vuke_amd.Engine(0, layout=layout, batch=256, blocks_per_cu=2)
info = e.info()

# ---- config 3: 1 000 000 targets = 999 000 decoys + 1 000 planted (seed 7919*j, sub j%3, kind cycling)
spec = ffi.spec(ffi.T_SHA256)
N_IN = 30_000_000
planted = [(7919 * j, j % 3, j % 3) for j in range(1000)]
rows = e.transform_batch(spec, [s for s, _, _ in planted])
keys = [rows[i][sub] for i, (_, sub, _) in enumerate(planted)]
_, hc, hu = e.derive_batch(keys)
h160s, kinds = [], []
for (seed, sub, kind), c, u in zip(planted, hc, hu):
    h160s.append(u if kind == 1 else c)
    kinds.append(ffi.TARGET_P2WPKH if kind == 2 else ffi.TARGET_P2PKH)
for j in range(999_000):
    h160s.append(hashlib.sha256(b"decoy" + struct.pack("<Q", j)).digest()[:20])
    kinds.append(j & 1)
t = time.perf_counter(); e.targets_set(h160s, kinds); t_set = time.perf_counter() - t
e.sweep(spec, 0, 2**20, flags=3)
t = time.perf_counter(); st, hits = e.sweep(spec, 0, N_IN - 1, flags=3, hit_cap=4096); dt = time.perf_counter() - t
exp = sum(1 for s, _, _ in planted if s < N_IN)
print(json.dumps({"config": "3: scan sha256 vs 1M targets (both hash160s, 3 keys/input)", "inputs": N_IN, "keys": st["keys"],
                  "matches": st["matches"], "expected_matches": exp, "keys_per_s": st["keys"] / dt, "kernel_keys_per_s": st["keys"] / st["kernel_seconds"],
                  "targets_set_s": t_set, "table": info}))
assert st["matches"] == exp
e.targets_set([], [])

# ---- config 4 slice: all LCG variants over 2^28 seeds of the 2^31 space (this GPU's share at 8 GPUs)
spec = ffi.spec(ffi.T_LCG)
t = time.perf_counter(); st, _ = e.sweep(spec, 3 << 28, (4 << 28) - 1, flags=1); dt = time.perf_counter() - t
print(json.dumps({"config": "4 (one of 8 shares): lcg all variants x 2^28 seeds, compressed", "keys": st["keys"], "keys_per_s": st["keys"] / dt,
                  "seconds": dt}))

# ---- per-transform rates, compressed only, 2^27 inputs
for name, sp in (("milksad", ffi.spec(ffi.T_MILKSAD)), ("mt64", ffi.spec(ffi.T_MT64)), ("xorshift", ffi.spec(ffi.T_XORSHIFT)),
                 ("sha256_chain", ffi.spec(ffi.T_SHA256_CHAIN))):
    n = (1 << 27) // e._L.vk_slots_per_input(__import__("ctypes").byref(sp))
    st, _ = e.sweep(sp, 0, n - 1, flags=1)
    print(json.dumps({"transform": name, "keys": st["keys"], "kernel_keys_per_s": st["keys"] / st["kernel_seconds"]}))
