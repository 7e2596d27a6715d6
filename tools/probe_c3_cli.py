"""BASELINE config 3 through the CLI (`scan --transform=sha256` of 10^8 inputs against a 1 000 000-address text file): wall time of
the run and of its parts as `VUKE_HIP_TIMING=1` reports them."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_workloads as wl
from oracle import vko
vko.build()
fx = wl.load_golden("bench_c3.json")
lines = []
for p in fx["planted"]:
    h = bytes.fromhex(p["h160"])
    lines.append(vko.segwit_encode("bc", 0, h) if p["addr_kind"] == 2 else vko.base58check_encode(b"\0" + h))
for h, kind in wl.c3_decoys():
    lines.append(vko.segwit_encode("bc", 0, h) if kind == 1 else vko.base58check_encode(b"\0" + h))
path = "/dev/shm/c3_targets.txt"
open(path, "w").write("\n".join(lines) + "\n")
exe = os.path.join(ROOT, "vuke_amd", "lib", "vuke-hip")
for i in range(3):
    t = time.perf_counter()
    r = subprocess.run([exe, "scan", "--transform", "sha256", "--targets", path, "range", "--start", "0", "--end", str(wl.C3_INPUTS - 1)],
                       capture_output=True, text=True, env=dict(os.environ, VUKE_HIP_TIMING="1"))
    dt = time.perf_counter() - t
    tl = [l for l in r.stderr.strip().splitlines() if l.startswith('[timing]') or l.startswith('Done')]
    print(f"cli run {i}: {dt:.3f} s  hits {r.stdout.count('========== HIT')}  stderr: {tl}", flush=True)
os.remove(path)
