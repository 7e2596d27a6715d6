import sys, os
sys.path.insert(0, "/root/repo")
import vuke_amd
from vuke_amd import _ffi as ffi
algo = int(os.environ.get("VK_ALGO", "0"))
batch = int(os.environ.get("VK_BATCH", "256"))
e = vuke_amd.Engine(0, layout=1, batch=batch, blocks_per_cu=int(os.environ.get("VK_BLOCKS", "2")), algo=algo)
sp = ffi.spec(ffi.T_MILKSAD)
e.sweep(sp, 0, 2**26 - 1)
for rep in range(3):
    tot = 0.0; n = 0
    for b in range(8):
        st, _ = e.sweep(sp, b << 28, ((b + 1) << 28) - 1, flags=1)
        tot += st["kernel_seconds"]; n += st["keys"]
    print(os.environ.get("VUKE_HIP_LIB", "default").split("/")[-1], "algo", algo, "batch", batch, "%.1f M keys/s (kernel)" % (n / tot / 1e6), flush=True)
