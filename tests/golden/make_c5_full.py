#!/usr/bin/env python3
"""Generate tests/golden/bench_c5_full.json: the CPU oracle over BASELINE config 5 at its stated size -- all 100 000 synthetic
MultiBit-HD mnemonics and all 100 000 Electrum v1 seeds of bench_workloads.py, 20 keys each: key count and the digest of
both hash160s of every key (what bench.py --config 5m / 5e sweeps in one step).  Minutes on 16 cores.

    python tests/golden/make_c5_full.py [--threads 16]
"""
import argparse
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import vko  # noqa: E402
import bench_workloads as wl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(HERE, "bench_c5_full.json"))
    args = ap.parse_args()
    words = vko.Wordlist(wl.c5_wordlist())
    mn, es = wl.c5_mnemonics(), wl.c5_electrum_seeds()

    def h160s(keys):
        out = []
        for k in keys:
            c, u = vko.derive_h160(k)
            out += [c, u]
        return out

    def mb(i):
        r = vko.multibit_keys(words, mn[i])
        return h160s([k for _, k in r[2]]) if r else []

    def el(i):
        return h160s([k for _, k in vko.electrum_keys(es[i])[1]])

    out = {"what": "config 5 at size: the oracle over all 100 000 inputs of each chain; digest over both hash160s of every key",
           "generator": "tests/golden/make_c5_full.py"}
    with ThreadPoolExecutor(args.threads) as ex:
        for name, fn, n in (("multibit", mb, len(mn)), ("electrum", el, len(es))):
            parts, keys = [], 0
            for lo in range(0, n, 5000):
                hs = [h for r in ex.map(fn, range(lo, min(n, lo + 5000))) for h in r]
                keys += len(hs) // 2
                parts.append(wl.digest_of_h160s(hs))
                print(name, lo, flush=True)
            out[name] = {"inputs": n, "keys": keys, "flags": 3, "digest": wl.hexd(wl.fold_digests(parts))}
    json.dump(out, open(args.out, "w"), indent=0)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
