"""Where the wall time of one vk_sweep call goes beyond its kernel (run on the GPU box)."""
import sys, time
sys.path.insert(0, "/root/repo")
import ctypes as C
import vuke_amd
from vuke_amd import _ffi as ffi
e = vuke_amd.Engine(0, layout=1, batch=256, blocks_per_cu=2)
sp = ffi.spec(ffi.T_MILKSAD)
e.sweep(sp, 0, 2**28 - 1)
L = e._L
for lg in (20, 24, 26, 28, 28, 28, 30):
    hits = (ffi.Hit * 1024)()
    st = ffi.SweepStats()
    t0 = time.perf_counter()
    rc = L.vk_sweep(e._e, C.byref(sp), 0, 2**lg - 1, 1, hits, 1024, C.byref(st))
    t1 = time.perf_counter()
    print(lg, "wall %.3f ms kernel %.3f ms gap %.3f ms launches %d" % ((t1 - t0) * 1e3, st.kernel_seconds * 1e3, (t1 - t0 - st.kernel_seconds) * 1e3, st.launches), flush=True)
