"""Sweep rate per transform kind on the bench geometry (signed9 table, batch 256): 2^26 inputs each, compressed hash160 only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi
e = vuke_amd.Engine(0, layout=1, batch=256, blocks_per_cu=2)
for name, spec, n in (("milksad", ffi.spec(ffi.T_MILKSAD), 1 << 27), ("mt64", ffi.spec(ffi.T_MT64), 1 << 27), ("sha256", ffi.spec(ffi.T_SHA256), 1 << 25),
                      ("lcg", ffi.spec(ffi.T_LCG), 1 << 25), ("xorshift", ffi.spec(ffi.T_XORSHIFT), 1 << 25), ("md5", ffi.spec(ffi.T_MD5), 1 << 27),
                      ("double_sha256", ffi.spec(ffi.T_DOUBLE_SHA256), 1 << 25), ("direct", ffi.spec(ffi.T_DIRECT), 1 << 25),
                      ("sha256_chain (4 variants x depth 10, staged)", ffi.spec(ffi.T_SHA256_CHAIN), 1 << 21), ("armory (staged)", ffi.spec(ffi.T_ARMORY), 1 << 22)):
    e.sweep(spec, 0, 1 << 18, flags=1)
    st, _ = e.sweep(spec, 1 << 30, (1 << 30) + n - 1, flags=1)
    print(f"{name:48s} {st['keys'] / st['kernel_seconds'] / 1e9:6.3f} G keys/s  ({st['keys']} keys, {st['launches']} launches, {st['kernel_seconds'] * 1e3:.1f} ms)", flush=True)
