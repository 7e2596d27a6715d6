"""Chain ring probe (GPU box): time of one vk_transform_strings call per config-5 chain for worker counts, chunk sizes and input
counts, through the vk_debug_ring_knobs hook of the test build (libvuke_hip_hooks.so)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi
import bench_workloads as wl

eng = vuke_amd.Engine(0, hooks=True)
eng.bip39_wordlist_set(wl.c5_wordlist())
which = sys.argv[1] if len(sys.argv) > 1 else "m"
lines_all = wl.c5_mnemonics(262144) if which == "m" else wl.c5_electrum_seeds(262144)
spec = ffi.spec(ffi.T_MULTIBIT if which == "m" else ffi.T_ELECTRUM, 0, 0, 20)

def run(n, wps=None, chunk=None, workers=None):
    eng.debug_ring_knobs(chunk_iters=chunk or 0, waves_per_simd=wps or 0, workers=workers or 0)
    lines = lines_all[:n]
    best = 1e9
    for _ in range(2):
        _, _, dt = eng.transform_strings(spec, lines, raw=True)
        best = min(best, dt)
    print(f"{which} n={n:7d} wps={wps} chunk={chunk} workers={workers}: {1e3 * best:8.1f} ms  {n / best / 1e3:8.1f} k inputs/s", flush=True)

run(4096)
for workers in (1024, 1280, 1563, 1792, 2048):
    run(100000, wps=2, workers=workers)
for chunk in ((8, 16, 32, 64, 128) if which == "m" else (50, 100, 250, 500, 2000)):
    run(100000, wps=1, chunk=chunk)
    run(100000, wps=2, workers=2048, chunk=chunk)
for n in (65536, 131072, 262144):
    run(n, wps=1)
    run(n)
