"""`generate --transform=sha256 --storage DIR wordlist` of 1 000 000 passphrases (console + 19-column Parquet, zstd): columns from the
device (vk_derive_columns; labels from the host) against the record path (VUKE_HIP_RECORD_PATH=1)."""
import hashlib, os, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "vuke_amd", "lib", "vuke-hip")
N = 1_000_000
tmp = "/dev/shm/probe_wls"
os.makedirs(tmp, exist_ok=True)
try:
    with open(tmp + "/w.txt", "w") as f:
        f.write("".join(f"passphrase-{i:07d}-{hashlib.md5(str(i).encode()).hexdigest()[:i % 17]}\n" for i in range(N)))
    for mode in ("device", "record"):
        env = dict(os.environ, VUKE_HIP_TIMING="1")
        if mode == "record":
            env["VUKE_HIP_RECORD_PATH"] = "1"
        best = None
        for i in range(2):
            shutil.rmtree(tmp + "/store", ignore_errors=True)
            t = time.perf_counter()
            r = subprocess.run([exe, "generate", "--transform", "sha256", "-o", tmp + "/out.txt", "--storage", tmp + "/store", "wordlist", "--file", tmp + "/w.txt"],
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        store = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(tmp + "/store") for f in fs)
        done = [l for l in r.stderr.decode().splitlines() if l.startswith("Done.")]
        print(f"{mode:7s}: best of 2 {best:.3f} s  text {os.path.getsize(tmp + '/out.txt') / 1e6:.1f} MB  parquet {store / 1e6:.1f} MB  rc {r.returncode} {done}", flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
