"""`generate --transform=sha256 wordlist` of 2 000 000 passphrases through the CLI (stdout to /dev/shm): the compact lines made on the
device (vk_generate_lines_strings) against the record path (VUKE_HIP_RECORD_PATH=1: keys and hash160s back to the host, hex /
Base58Check / line assembly on the host threads), same binary, outputs compared."""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "vuke_amd", "lib", "vuke-hip")
N = 2_000_000
wl = "/dev/shm/probe_wl.txt"
with open(wl, "w") as f:
    f.write("".join(f"passphrase-{i:07d}-{hashlib.md5(str(i).encode()).hexdigest()[:i % 17]}\n" for i in range(N)))
try:
    digests = {}
    for transforms in (["sha256"], ["sha256", "md5", "sha256_chain"], ["armory"]):
        for mode in ("text", "record"):
            env = dict(os.environ)
            if mode == "record":
                env["VUKE_HIP_RECORD_PATH"] = "1"
            best = None
            for i in range(2):
                t = time.perf_counter()
                with open("/dev/shm/probe_wl.out", "wb") as f:
                    r = subprocess.run([exe, "generate", "--transform", *transforms, "wordlist", "--file", wl], stdout=f, stderr=subprocess.PIPE, env=env)
                dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            h = hashlib.sha256(open("/dev/shm/probe_wl.out", "rb").read()).hexdigest()[:16]
            digests.setdefault(tuple(transforms), set()).add(h)
            print(f"{'+'.join(transforms):28s} {mode:6s}: best of 2 {best:.3f} s  out {os.path.getsize('/dev/shm/probe_wl.out')} B sha256 {h}  [{r.stderr.decode().strip().splitlines()[-1]}]", flush=True)
    print("outputs identical between the two paths:", all(len(v) == 1 for v in digests.values()))
finally:
    for p in (wl, "/dev/shm/probe_wl.out"):
        if os.path.exists(p):
            os.remove(p)
