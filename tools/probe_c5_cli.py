"""BASELINE config 5 through the CLI in generate mode: 100 000 synthetic mnemonics (multibit) and 100 000 Electrum seeds from a
wordlist file, 20 keys each, compact console output to /dev/shm -- lines made on the device (labels: the shortened forms the
reference prints, made on the host once per input) against the record path (VUKE_HIP_RECORD_PATH=1), outputs compared."""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_workloads as wl
exe = os.path.join(ROOT, "vuke_amd", "lib", "vuke-hip")
tmp = "/dev/shm/probe_c5"
os.makedirs(tmp, exist_ok=True)
try:
    open(tmp + "/bip39.txt", "w").write("\n".join(wl.c5_wordlist()) + "\n")
    open(tmp + "/m.txt", "wb").write(b"\n".join(wl.c5_mnemonics()) + b"\n")
    open(tmp + "/e.txt", "wb").write(b"\n".join(wl.c5_electrum_seeds()) + b"\n")
    for name, args in (("multibit", ["--transform", "multibit", "--bip39-wordlist", tmp + "/bip39.txt", "wordlist", "--file", tmp + "/m.txt"]),
                       ("electrum", ["--transform", "electrum", "wordlist", "--file", tmp + "/e.txt"])):
        digests = set()
        for mode in ("text", "record"):
            env = dict(os.environ, VUKE_HIP_TIMING="1")
            if mode == "record":
                env["VUKE_HIP_RECORD_PATH"] = "1"
            best = None
            for i in range(2):
                t = time.perf_counter()
                r = subprocess.run([exe, "generate", "-o", tmp + "/out.txt"] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
                dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            h = hashlib.sha256(open(tmp + "/out.txt", "rb").read()).hexdigest()[:16]
            digests.add(h)
            done = [l for l in r.stderr.decode().splitlines() if l.startswith("Done.")]
            print(f"{name:9s} {mode:6s}: best of 2 {best:.3f} s  out {os.path.getsize(tmp + '/out.txt')} B sha256 {h}  {done}", flush=True)
        print(f"{name}: outputs identical between the two paths: {len(digests) == 1}", flush=True)
finally:
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
