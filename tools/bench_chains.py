#!/usr/bin/env python3
"""Config 5 on one GPU: throughput of the MultiBit-HD and Electrum v1 key-chain kernels
(inputs/s and keys/s over the C-ABI call: host-to-device copy, kernel, device-to-host copy), with canonical op rates per SURVEY 8(d)."""
import json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
e = vuke_amd.Engine(0)
rnd = random.Random(5489)
# synthetic inputs only: a pseudo-random 2048-word list and mnemonics / hex seeds drawn from a fixed RNG
words = sorted({"".join(rnd.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rnd.randrange(3, 9))) for _ in range(4000)})[:2048]
e.bip39_wordlist_set(words)
mn = [" ".join(rnd.choice(words) for _ in range(12)).encode() for _ in range(n)]
spec = ffi.spec(ffi.T_MULTIBIT, 0, 0, 20)
e.transform_strings(spec, mn[:1024])
_, valid, dt = e.transform_strings(spec, mn, raw=True)   # dt = inside vk_transform_strings: H2D + kernel + D2H
keys = sum(valid)
ops = 2 * 2048 * 2 * 5520 + 40 * 213000
print(json.dumps({"chain": "multibit", "inputs": n, "seconds": dt, "inputs_per_s": n / dt, "keys_per_s": keys / dt,
                  "canonical_tops": n / dt * ops / 1e12}))
seeds = [("%032x" % rnd.getrandbits(128)).encode() for _ in range(n)]
spec = ffi.spec(ffi.T_ELECTRUM, 0, 0, 20)
e.transform_strings(spec, seeds[:1024])
_, valid, dt = e.transform_strings(spec, seeds, raw=True)
keys = sum(valid)
ops = 200000 * 2168 + 21 * 213000
print(json.dumps({"chain": "electrum", "inputs": n, "seconds": dt, "inputs_per_s": n / dt, "keys_per_s": keys / dt,
                  "canonical_tops": n / dt * ops / 1e12}))
# Bitimage with a passphrase wordlist: 64 files x n/64 passphrases, default path, one key each (one PBKDF2 per item)
import time
ents = [rnd.randbytes(32) for _ in range(64)]
pws = [("pw%07d" % i).encode() for i in range(n // 64)]
e.bitimage_keys(ents[:4], pws[:256])
t0 = time.perf_counter()
res = e.bitimage_keys(ents, pws)
dt = time.perf_counter() - t0   # includes the python-side unpacking of the result
items = len(ents) * len(pws)
ops = 2048 * 2 * 5520 + 3 * 213000
print(json.dumps({"chain": "bitimage", "items": items, "seconds_incl_python_unpack": dt, "items_per_s": items / dt,
                  "canonical_tops": items / dt * ops / 1e12}))
