import sys, time
sys.path.insert(0, "/root/repo")
t0 = time.perf_counter()
import vuke_amd
from vuke_amd import _ffi as ffi
t1 = time.perf_counter()
e = vuke_amd.Engine(0, layout=1)
t2 = time.perf_counter()
print("import %.2f s, engine create %.2f s (table build %.3f s)" % (t1 - t0, t2 - t1, e.info()["table_build_seconds"]))
sp = ffi.spec(ffi.T_MILKSAD)
for flags in (1, 3):
    t = time.perf_counter()
    n = 0
    for b in range(4):
        st, _ = e.sweep(sp, b << 30, ((b + 1) << 30) - 1, flags=flags)
        n += st["keys"]
    dt = time.perf_counter() - t
    print("flags %d: 2^32 in %.2f s = %.0f M keys/s" % (flags, dt, n / dt / 1e6))
t = time.perf_counter(); e.close(); print("destroy %.2f s" % (time.perf_counter() - t))
