import sys, os, time
sys.path.insert(0, os.getcwd())
import vuke_amd
for kw in (dict(window_bits=16), dict(), dict(window_bits=26), dict(layout=1, batch=256, blocks_per_cu=2)):
    t = time.perf_counter(); e = vuke_amd.Engine(0, **kw); dt = time.perf_counter() - t
    print(kw, "create %.3f s  table build %.4f s" % (dt, e.info()["table_build_seconds"]), flush=True); e.close()
