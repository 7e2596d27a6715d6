"""Does a sweep on one engine make progress while another thread is inside the slow hipMalloc of a big table (VRAM being scrubbed after
a previous release)?  Decides whether a 'start on the small table, switch to the big one when it is ready' engine is worth building."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi

big = vuke_amd.Engine(0, layout=1, batch=256, blocks_per_cu=2)   # 120 GB
big.close()                                                      # released: the next big allocation waits for the scrub
t0 = time.perf_counter()
small = vuke_amd.Engine(0)                                       # 11.8 GB
print(f"small engine created in {time.perf_counter() - t0:.3f} s", flush=True)
res = {}

def make_big():
    t = time.perf_counter()
    e = vuke_amd.Engine(0, layout=1, batch=256, blocks_per_cu=2)
    res["big_s"] = time.perf_counter() - t
    res["eng"] = e

th = threading.Thread(target=make_big)
t1 = time.perf_counter()
th.start()
sp = ffi.spec(ffi.T_MILKSAD)
i = 0
while th.is_alive():
    t = time.perf_counter()
    st, _ = small.sweep(sp, (i % 32) << 26, (((i % 32) + 1) << 26) - 1, flags=1)
    print(f"  sweep {i} of 2^26 seeds on the small table: {1e3 * (time.perf_counter() - t):7.1f} ms wall, {1e3 * st['kernel_seconds']:7.1f} ms kernel, at t = {time.perf_counter() - t1:.2f} s", flush=True)
    i += 1
th.join()
print(f"big engine created in {res['big_s']:.3f} s while {i} sweeps ran; alone a 2^26 sweep takes:", flush=True)
t = time.perf_counter(); st, _ = small.sweep(sp, 0, (1 << 26) - 1, flags=1); print(f"  {1e3 * (time.perf_counter() - t):.1f} ms wall, {1e3 * st['kernel_seconds']:.1f} ms kernel")
