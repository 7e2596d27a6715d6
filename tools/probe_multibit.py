import sys, os, time
sys.path.insert(0, os.getcwd())
import vuke_amd
from vuke_amd import _ffi as ffi
import bench_workloads as wl
e = vuke_amd.Engine(0)
e.bip39_wordlist_set(wl.c5_wordlist())
for n in (100000, 262144):
    mn = wl.c5_mnemonics(n)
    packed = ffi.pack_strings(mn)
    spec = ffi.spec(ffi.T_MULTIBIT, 0, 0, 20)
    e.scan_strings(spec, None, flags=3, packed=ffi.pack_strings(mn[:1024]))
    st, _ = e.scan_strings(spec, None, flags=3, packed=packed)
    print(os.environ.get("VUKE_HIP_LIB","default").split("/")[-1], "multibit n", n, "kernel_s", round(st["kernel_seconds"],4), "mnemonics/s", round(n/st["kernel_seconds"]))
