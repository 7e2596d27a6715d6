#!/usr/bin/env python3
"""Generate tests/golden/soak_windows.json: one large oracle-swept window per transform kind that has no full-size golden of
its own (configs 2-5 have: milksad, lcg, sha256, multibit, electrum) -- mt64, the four xorshift generators, sha256_chain
(all variants x depth 10 x both seed forms), double_sha256, md5, direct, armory -- both hash160s of every key, around seed
ceilings where the transform has one.  ~5 x 10^8 oracle keys: a few minutes on 16 cores.

    python tests/golden/make_soak_windows.py [--threads 16]
"""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import vko  # noqa: E402
import bench_workloads as wl  # noqa: E402

WINDOWS = [  # name, kind, variant, endian, depth, lo, inputs
    ("mt64", vko.T_MT64, vko.VARIANT_ALL, 0, 10, 2**32 - 2**25, 2**26),
    ("xorshift", vko.T_XORSHIFT, vko.VARIANT_ALL, 0, 10, 2**63 - 2**23, 2**24),
    ("sha256_chain", vko.T_SHA256_CHAIN, vko.VARIANT_ALL, 0, 10, 2**32 - 2**19, 2**20),
    ("double_sha256", vko.T_DOUBLE_SHA256, vko.VARIANT_ALL, 0, 10, 10**12, 2**24),
    ("md5", vko.T_MD5, vko.VARIANT_ALL, 0, 10, 10**15, 2**24),
    ("direct", vko.T_DIRECT, vko.VARIANT_ALL, 0, 10, 2**64 - 2**24, 2**24),
    ("lcg:le", vko.T_LCG, vko.VARIANT_ALL, 1, 10, 2**31 - 2**23, 2**24),
    ("armory", vko.T_ARMORY, vko.VARIANT_ALL, 0, 10, 123456789, 2**18),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(HERE, "soak_windows.json"))
    args = ap.parse_args()
    out = []
    for name, kind, variant, endian, depth, lo, n in WINDOWS:
        sp = vko.spec(kind, variant, endian, depth)
        st, _ = vko.sweep(sp, lo, lo + n - 1, flags=3 | vko.F_BATCHED, nthreads=args.threads)
        out.append({"name": name, "spec": [kind, variant, endian, depth], "lo": lo, "hi": lo + n - 1, "flags": 3, "inputs": st["inputs"],
                    "keys": st["keys"], "digest": wl.hexd(st["digest"])})
        print(name, st["keys"], flush=True)
    json.dump({"what": "large oracle-swept windows of the transform kinds without a full-size golden", "generator": "tests/golden/make_soak_windows.py",
               "windows": out}, open(args.out, "w"), indent=0)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
