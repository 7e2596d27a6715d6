"""BASELINE config 2 through the CLI: `vuke-hip scan --transform milksad --targets T range --start 0 --end 4294967295` with the 1026
planted addresses of the fixture -- the full 2^32 Milksad sweep as a user runs it, wall time of the whole process."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_workloads as wl
from oracle import vko
vko.build()
pl = wl.load_golden("milksad_planted.json")["planted"]
path = "/dev/shm/c2_targets.txt"
open(path, "w").write("\n".join(vko.base58check_encode(b"\0" + bytes.fromhex(p["h160_c"])) for p in pl) + "\n")
exe = os.path.join(ROOT, "vuke_amd", "lib", "vuke-hip")
for extra in ([], ["--devices", "0"]):
    for i in range(2):
        t = time.perf_counter()
        r = subprocess.run([exe, "scan", "--transform", "milksad", "--targets", path, *extra, "range", "--start", "0", "--end", "4294967295"],
                           capture_output=True, text=True, env=dict(os.environ, VUKE_HIP_TIMING="1"))
        dt = time.perf_counter() - t
        tl = [l for l in r.stderr.strip().splitlines() if l.startswith('[timing]') or l.startswith('Done')]
        print(f"cli {' '.join(extra)} run {i}: {dt:.3f} s  hits {r.stdout.count('========== HIT')}  stderr: {tl}", flush=True)
os.remove(path)
