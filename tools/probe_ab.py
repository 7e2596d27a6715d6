"""Kernel-time A/B of builds of libvuke_hip.so on one box (VUKE_HIP_LIB selects the build): 3 x 8 launches of 2^28
Milksad seeds on the bench geometry, after checking the build against the oracle's 2^20-seed window fixture."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vuke_amd
from vuke_amd import _ffi as ffi
import bench_workloads as wl
batch = int(os.environ.get("VK_BATCH", "256"))
e = vuke_amd.Engine(0, layout=int(os.environ.get("VK_LAYOUT", "1")), batch=batch, blocks_per_cu=int(os.environ.get("VK_BLOCKS", "2")))
sp = ffi.spec(ffi.T_MILKSAD)
name = os.environ.get("VUKE_HIP_LIB", "default").split("/")[-1]
w = wl.load_golden("digest_windows.json")[0]
st, _ = e.sweep(sp, w["lo"], w["hi"], flags=1)
ok = wl.hexd(st["digest"]) == w["digest"]
e.sweep(sp, 0, 2**26 - 1)
dg = None
for rep in range(int(os.environ.get("VK_REPS", "3"))):
    tot = 0.0; n = 0
    for b in range(8):
        st, _ = e.sweep(sp, b << 28, ((b + 1) << 28) - 1, flags=1)
        tot += st["kernel_seconds"]; n += st["keys"]
        if b == 0:
            dg = wl.hexd(st["digest"])[0]
    print(name, "batch", batch, "oracle-window", "OK" if ok else "MISMATCH", "block0", dg, "%.1f M keys/s (kernel)" % (n / tot / 1e6), flush=True)
