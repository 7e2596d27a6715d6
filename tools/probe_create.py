import time, sys, os
sys.path.insert(0, os.getcwd())
import vuke_amd
for kw in (dict(), dict(window_bits=16), dict(window_bits=16, batch=32), dict(window_bits=20)):
    t = time.perf_counter(); e = vuke_amd.Engine(0, **kw); dt = time.perf_counter() - t
    print(kw, "create %.3f s" % dt, e.info()["table_build_seconds"]); e.close()
