"""Engine creation per table geometry (first engine of the process pays the HIP runtime start too; every geometry is created
twice), and what a small generate job costs with each: BASELINE config 1's 3 000 003 keys through vk_generate_lines."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi
sp = ffi.spec(ffi.T_SHA256)
for kw in (dict(window_bits=16), dict(window_bits=16), dict(window_bits=20), dict(), dict(window_bits=26), dict(layout=1, batch=256, blocks_per_cu=2)):
    for rep in range(2):
        t = time.perf_counter(); e = vuke_amd.Engine(0, **kw); dt = time.perf_counter() - t
        t = time.perf_counter()
        keys = 0
        for lo in range(0, 1000001, 1 << 19):
            text, st = e.generate_lines(sp, lo, min(lo + (1 << 19) - 1, 1000000), "sha256", [""] * 3)
            keys += st["keys"]
        dg = time.perf_counter() - t
        t = time.perf_counter(); e.close(); dc = time.perf_counter() - t
        print("%-48s create %.3f s | config-1 lines %.3f s (%d keys) | destroy %.3f s" % (kw, dt, dg, keys, dc), flush=True)
