"""scan --transform=sha256 of a 20 000 000-line wordlist (the brain-wallet use of the tool): the device needs milliseconds per
million lines, so this measures the host's line reader and batching."""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "vuke_amd", "lib", "vuke-hip")
N = 20_000_000
wl, tf = "/dev/shm/probe_scan_wl.txt", "/dev/shm/probe_scan_targets.txt"
try:
    t = time.perf_counter()
    with open(wl, "w") as f:
        for base in range(0, N, 1_000_000):
            f.write("".join(f"pass{i:08d}word{i % 977}\n" for i in range(base, base + 1_000_000)))
    print(f"wordlist written: {os.path.getsize(wl) / 1e6:.0f} MB in {time.perf_counter() - t:.1f} s", flush=True)
    open(tf, "w").write("1JwSSubhmg6iPtRjtyqhUYYH7bZg3Lfy1T\n1C7zdTfnkzmr13HfA2vNm5SJYRK6nEKyq8\n")
    for extra in ([], ["--transform", "sha256", "md5", "double_sha256"]):
        args = [exe, "scan"] + (extra or ["--transform", "sha256"]) + ["--targets", tf, "wordlist", "--file", wl]
        for i in range(2):
            t = time.perf_counter()
            r = subprocess.run(args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, VUKE_HIP_TIMING="1"))
            dt = time.perf_counter() - t
            tail = [l for l in r.stderr.decode().splitlines() if l.startswith("Done.") or l.startswith("[timing]")]
            print(f"{' '.join(args[2:-5])}: run {i}: {dt:.3f} s = {N / dt / 1e6:.1f} M lines/s  {tail}", flush=True)
finally:
    for p in (wl, tf):
        if os.path.exists(p):
            os.remove(p)
