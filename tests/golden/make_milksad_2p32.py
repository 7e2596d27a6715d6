#!/usr/bin/env python3
"""Generate tests/golden/milksad_2p32.json: the CPU oracle swept over the WHOLE Milksad seed space.

BASELINE config 2 at its stated size, computed once by the oracle (never by the device):
  * digest of the compressed hash160 of every seed 0..2^32-1 (DESIGN.md "sweep digest"), as 256 windows of
    2^24 seeds, which fold into the 16 blocks of 2^28 seeds bench.py steps over and into the full-space value;
  * (milksad_planted.json) the planted target set of bench.py / tests (1024 seeds (0x9E3779B1*j + 12345) mod 2^32,
    plus seeds 0 and 2^32-1): seed, private key and compressed hash160 from vko_milksad_key + vko_derive (the
    unbatched path).
The sweep itself uses VKO_F_BATCHED (16-bit windows + batch inversion; tests/test_oracle_derive.py holds it to
the plain double-and-add path).  Hours of CPU: progress is checkpointed per window in milksad_2p32.partial.jsonl
(git-ignored) and the run resumes from it.

    python tests/golden/make_milksad_2p32.py [--threads 6] [--windows 0:256]

Reference behaviour restated by the oracle: src/transform/milksad.rs:17-29, src/derive.rs:78-118.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import vko  # noqa: E402

PARTIAL = os.environ.get("VK_PARTIAL", os.path.join(HERE, "milksad_2p32.partial.jsonl"))  # several machines may share the work
OUT = os.path.join(HERE, "milksad_2p32.json")
PLANTED = os.path.join(HERE, "milksad_planted.json")
WLOG2 = 24
NWIN = 1 << (32 - WLOG2)
MASK = 2**64 - 1


def fold(parts):
    d = [0, 0, 0, 0]
    for p in parts:
        for i in range(3):
            d[i] ^= p[i]
        d[3] = (d[3] + p[3]) & MASK
    return d


def planted_seeds():
    return sorted({(0x9E3779B1 * j + 12345) & 0xFFFFFFFF for j in range(1024)} | {0, 2**32 - 1})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--windows", default=f"0:{NWIN}")
    ap.add_argument("--planted-only", action="store_true", help="write only milksad_planted.json (seconds)")
    args = ap.parse_args()
    sp = vko.spec(vko.T_MILKSAD)
    planted = []
    for s in planted_seeds():
        key = vko.transform_apply(sp, s)[0][1]
        d = vko.derive(key)  # the unbatched KeyDeriver restatement
        planted.append({"seed": s, "privkey": bytes(d.raw).hex(), "h160_c": bytes(d.h160_c).hex()})
    json.dump({"what": "planted targets of the Milksad sweep (bench.py, tests): seed -> key -> compressed hash160, from the CPU oracle",
               "generator": "tests/golden/make_milksad_2p32.py", "rule": "seeds (0x9E3779B1*j + 12345) mod 2^32 for j < 1024, plus 0 and 2^32-1",
               "planted": planted}, open(PLANTED, "w"), indent=0)
    print("wrote", PLANTED)
    if args.planted_only:
        return
    w0, w1 = (int(x) for x in args.windows.split(":"))
    done = {}
    if os.path.exists(PARTIAL):
        for line in open(PARTIAL):
            r = json.loads(line)
            done[r["window"]] = r
    for w in range(w0, w1):
        if w in done:
            continue
        lo = w << WLOG2
        t = time.time()
        st, _ = vko.sweep(sp, lo, lo + (1 << WLOG2) - 1, flags=vko.F_COMPRESSED | vko.F_BATCHED, nthreads=args.threads)
        assert st["keys"] == 1 << WLOG2
        r = {"window": w, "digest": [f"{d:016x}" for d in st["digest"]], "seconds": round(time.time() - t, 1)}
        with open(PARTIAL, "a") as f:
            f.write(json.dumps(r) + "\n")
        done[w] = r
        print(f"window {w}: {r['seconds']} s", flush=True)
    if len(done) < NWIN:
        print(f"{len(done)}/{NWIN} windows done; golden file not written yet")
        return
    wd = [[int(x, 16) for x in done[w]["digest"]] for w in range(NWIN)]
    blocks = [fold(wd[b * 16:(b + 1) * 16]) for b in range(16)]
    out = {
        "what": "CPU-oracle sweep of Milksad seeds 0..2^32-1, compressed hash160 digest (xor,xor,xor,sum) + planted targets",
        "generator": "tests/golden/make_milksad_2p32.py",
        "window_log2": WLOG2,
        "window_digests": [[f"{d:016x}" for d in x] for x in wd],
        "block_log2": 28,
        "block_digests": [[f"{d:016x}" for d in x] for x in blocks],
        "digest": [f"{d:016x}" for d in fold(wd)],
        "planted_file": "milksad_planted.json",
    }
    json.dump(out, open(OUT, "w"), indent=0)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
