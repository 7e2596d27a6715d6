import sys, time
sys.path.insert(0, "/root/repo")
from vuke_amd import _ffi as ffi
import vuke_amd
lay = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for blocks, batch in ((2, 256), (2, 384), (4, 256), (1, 512), (2, 512), (2, 320), (2, 300), (2, 288), (2, 272)):
    e = vuke_amd.Engine(0, layout=lay, batch=batch, blocks_per_cu=blocks) if lay else vuke_amd.Engine(0, batch=batch, blocks_per_cu=blocks)
    sp = ffi.spec(ffi.T_MILKSAD)
    e.sweep(sp, 0, 2**24 - 1)
    for lg in (26, 28):
        st, _ = e.sweep(sp, 0, 2**lg - 1)
        print(lay, blocks, batch, lg, "%.1f Mkeys/s" % (st["keys"] / st["kernel_seconds"] / 1e6), flush=True)
    e.close()
