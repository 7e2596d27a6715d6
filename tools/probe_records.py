"""BASELINE config 1's inputs (generate --transform=sha256 range 0..=1000000, 3 000 003 keys) through every sink of the CLI:
where the time of the full-record formats goes ("records" = VUKE_HIP_RECORD_PATH=1: keys and hash160s back to the host, every
field formatted there; otherwise the lines / the record's columns are made on the device).  Everything is written to /dev/shm."""
import os, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "vuke_amd", "lib", "vuke-hip")
base = ["generate", "--transform", "sha256"]
rng = ["range", "--start", "0", "--end", "1000000"]
out = "/dev/shm/probe_rec.out"
sdir = "/dev/shm/probe_rec_store"
cases = [
    ("compact, lines made on the device", base + ["-o", out] + rng, {}),
    ("compact, record path", base + ["-o", out] + rng, {"VUKE_HIP_RECORD_PATH": "1"}),
    ("verbose", base + ["-v", "-o", out] + rng, {}),
    ("compact + parquet (none), records", base + ["-o", out, "--storage", sdir, "--compression", "none"] + rng, {"VUKE_HIP_RECORD_PATH": "1"}),
    ("compact + parquet (none)", base + ["-o", out, "--storage", sdir, "--compression", "none"] + rng, {}),
    ("compact + parquet (zstd)", base + ["-o", out, "--storage", sdir] + rng, {}),
    ("compact + parquet (snappy)", base + ["-o", out, "--storage", sdir, "--compression", "snappy"] + rng, {}),
]
try:
    for name, args, env in cases:
        best = None
        for i in range(2):
            shutil.rmtree(sdir, ignore_errors=True)
            t = time.perf_counter()
            r = subprocess.run([exe] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, VUKE_HIP_TIMING="1", **env))
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        size = os.path.getsize(out) if os.path.exists(out) else 0
        store = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(sdir) for f in fs) if os.path.isdir(sdir) else 0
        print(f"{name:36s}: best of 2 {best:6.3f} s   text {size / 1e6:8.1f} MB   parquet {store / 1e6:8.1f} MB   rc {r.returncode}", flush=True)
        print("      " + " | ".join(l for l in r.stderr.decode().splitlines() if l.startswith("[timing]")), flush=True)
finally:
    shutil.rmtree(sdir, ignore_errors=True)
    if os.path.exists(out):
        os.remove(out)
