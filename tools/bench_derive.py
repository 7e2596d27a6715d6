#!/usr/bin/env python3
"""KeyDeriver::derive throughput through the C ABI (the measure of the reference's Criterion benches derive_addresses /
derive_batch_100, benches/derivation.rs:12-34): vk_derive_batch over host buffers, PCIe copies included."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi

e = vuke_amd.Engine(0, layout=1, batch=256, blocks_per_cu=2)
L = e._L
for n in (100, 1 << 16, 1 << 22, 1 << 24):
    keys = os.urandom(32 * n)
    norm, hc, hu = C.create_string_buffer(32 * n), C.create_string_buffer(20 * n), C.create_string_buffer(20 * n)
    L.vk_derive_batch(e._e, keys, n, norm, hc, hu)
    reps = 200 if n <= 100 else 3
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = L.vk_derive_batch(e._e, keys, n, norm, hc, hu)
    dt = (time.perf_counter() - t0) / reps
    assert rc == 0
    print(json.dumps({"bench": "derive_batch", "keys": n, "seconds_per_call": dt, "keys_per_s": n / dt,
                      "outputs": "normalised key + hash160 compressed + hash160 uncompressed (72 B/key back over PCIe)"}))
