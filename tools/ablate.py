#!/usr/bin/env python3
"""Timing ablation of the sweep kernel: what the transform and each hash160 cost (ns per key-lane)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi
W = int(sys.argv[1]) if len(sys.argv) > 1 else 24
e = vuke_amd.Engine(0, window_bits=W, batch=64)
n = 1 << 27
def t(spec, flags):
    e.sweep(spec, 0, (1 << 22) - 1, flags=flags)
    st, _ = e.sweep(spec, 0, n - 1, flags=flags)
    return st["kernel_seconds"] / st["keys"] * 1e9, st["keys"]
rows = {}
for name, spec in (("milksad", ffi.spec(ffi.T_MILKSAD)), ("lcg:glibc", ffi.spec(ffi.T_LCG, 0)), ("mt64", ffi.spec(ffi.T_MT64)),
                   ("xorshift:64", ffi.spec(ffi.T_XORSHIFT, 0)), ("sha256", ffi.spec(ffi.T_SHA256))):
    for flags in (1, 2, 3):
        ns, keys = t(spec, flags)
        rows[(name, flags)] = ns
        print(f"{name:12s} flags={flags} {ns*1000:8.1f} ps/key  ({1e9/ns/1e6:7.1f} M keys/s)  keys={keys}")
m = rows
print("hash160(compressed)   ~", (m[("lcg:glibc", 3)] - m[("lcg:glibc", 2)]) * 1000, "ps/key")
print("hash160(uncompressed) ~", (m[("lcg:glibc", 3)] - m[("lcg:glibc", 1)]) * 1000, "ps/key")
print("MT19937 first-8       ~", (m[("milksad", 1)] - m[("lcg:glibc", 1)]) * 1000, "ps/key")
print("MT19937-64 first-4    ~", (m[("mt64", 1)] - m[("lcg:glibc", 1)]) * 1000, "ps/key")
print("sha256 transform      ~", (m[("sha256", 1)] - m[("lcg:glibc", 1)]) * 1000, "ps/key")
print("EC + rest (lcg, c)    ~", (m[("lcg:glibc", 1)] - (m[("lcg:glibc", 3)] - m[("lcg:glibc", 2)])) * 1000, "ps/key")
