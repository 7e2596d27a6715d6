"""Config 1 (generate --transform=sha256 range 0..=1000000, 3 000 003 compact lines): where the time goes.  CLI wall time
(stdout to /dev/shm), the device part alone (vk_generate over the same range through the C ABI), engine creation."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
exe = os.path.join(ROOT, "vuke_amd", "lib", "vuke-hip")
for i in range(3):
    t = time.perf_counter()
    with open("/dev/shm/c1.out", "wb") as f:
        r = subprocess.run([exe, "generate", "--transform", "sha256", "range", "--start", "0", "--end", "1000000"], stdout=f, stderr=subprocess.PIPE, env=dict(os.environ, VUKE_HIP_TIMING="1"))
    dt = time.perf_counter() - t
    print(f"cli run {i}: {dt:.3f} s  ({os.path.getsize('/dev/shm/c1.out')} bytes)  stderr tail: {r.stderr.decode().strip().splitlines()[-3:]}")
os.remove("/dev/shm/c1.out")
import vuke_amd
from vuke_amd import _ffi as ffi
t = time.perf_counter(); e = vuke_amd.Engine(0); print(f"engine create (W=24): {time.perf_counter() - t:.3f} s")
sp = ffi.spec(ffi.T_SHA256)
e.generate(sp, 0, 1000)
t = time.perf_counter(); g = e.generate(sp, 0, 1000000); dt = time.perf_counter() - t
print(f"vk_generate 0..1000000 through ctypes (one call, buffers allocated by python): {dt:.3f} s, kernel {g['kernel_seconds']:.4f} s, keys {g['keys']}")
