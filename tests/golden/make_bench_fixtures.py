#!/usr/bin/env python3
"""Generate tests/golden/bench_c3.json, bench_c4.json, bench_c5.json from the CPU oracle: the planted targets of
BASELINE configs 3, 4 and 5 (inputs defined in bench_workloads.py) and, per config, the oracle's
counters and digest over the warm-up window bench.py sweeps before its timed steps -- so that every expected value
bench.py and the -m gpu tests compare against comes from the oracle, never from the device.

    python tests/golden/make_bench_fixtures.py [--threads 2]
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


def stats(st):
    return {"inputs": st["inputs"], "keys": st["keys"], "digest": wl.hexd(st["digest"])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=2)
    args = ap.parse_args()
    T = args.threads

    # ---- config 3: sha256 transform, 1000 planted of mixed kinds
    sp = vko.spec(vko.T_SHA256)
    planted = []
    for seed, sub, kind in wl.c3_planted():
        d = vko.derive(vko.transform_key(sp, seed, sub))
        h = bytes(d.h160_u) if kind == 1 else bytes(d.h160_c)
        planted.append({"seed": seed, "sub_index": sub, "addr_kind": kind, "h160": h.hex(), "privkey": bytes(d.raw).hex()})
    st, _ = vko.sweep(sp, 0, wl.C3_WARMUP_INPUTS - 1, flags=3 | vko.F_BATCHED, nthreads=T)
    json.dump({"what": "config 3 (scan --transform=sha256 vs 1M targets): planted targets + oracle digest of the warm-up window",
               "generator": "tests/golden/make_bench_fixtures.py", "planted": planted,
               "warmup": {"lo": 0, "hi": wl.C3_WARMUP_INPUTS - 1, "flags": 3, **stats(st)}},
              open(os.path.join(HERE, "bench_c3.json"), "w"), indent=0)
    print("c3 done", flush=True)

    # ---- config 4: all LCG variants, 256 planted per variant
    sp = vko.spec(vko.T_LCG)
    planted = []
    for seed, v in wl.c4_planted():
        k = vko.transform_key(sp, seed, v)
        assert k is not None, (seed, v)
        d = vko.derive(k)
        planted.append({"seed": seed, "sub_index": v, "addr_kind": 0, "h160": bytes(d.h160_c).hex(), "privkey": bytes(d.raw).hex()})
    st, _ = vko.sweep(sp, 0, wl.C4_WARMUP_SEEDS - 1, flags=1 | vko.F_BATCHED, nthreads=T)
    json.dump({"what": "config 4 (all LCG variants x 2^31 seeds): planted targets + oracle digest of the warm-up window",
               "generator": "tests/golden/make_bench_fixtures.py", "planted": planted,
               "warmup": {"lo": 0, "hi": wl.C4_WARMUP_SEEDS - 1, "flags": 1, **stats(st)}},
              open(os.path.join(HERE, "bench_c4.json"), "w"), indent=0)
    print("c4 done", flush=True)

    # ---- config 5: MultiBit-HD mnemonics and Electrum v1 seeds, key 7 of every 1000th input planted
    words = vko.Wordlist(wl.c5_wordlist())
    mn = wl.c5_mnemonics()
    es = wl.c5_electrum_seeds()

    def mb(i):
        r = vko.multibit_keys(words, mn[i])
        return [k for _, k in r[2]] if r else []

    def el(i):
        return [k for _, k in vko.electrum_keys(es[i])[1]]

    out = {"what": "config 5 (MultiBit-HD / Electrum v1 chains, 100 000 synthetic inputs): planted targets + oracle digest of the warm-up inputs",
           "generator": "tests/golden/make_bench_fixtures.py"}
    with ThreadPoolExecutor(T) as ex:
        for name, fn, nwarm, use_u in (("multibit", mb, wl.C5_WARMUP_MULTIBIT, False), ("electrum", el, wl.C5_WARMUP_ELECTRUM, True)):
            planted = []
            pin = wl.c5_planted_inputs()
            for i, keys in zip(pin, ex.map(fn, pin)):
                assert len(keys) == 20, (name, i)
                d = vko.derive(keys[wl.C5_PLANT_INDEX])
                planted.append({"input": i, "sub_index": wl.C5_PLANT_INDEX, "addr_kind": 1 if use_u else 0,
                                "h160": (bytes(d.h160_u) if use_u else bytes(d.h160_c)).hex(), "privkey": bytes(d.raw).hex()})
            hs, nkeys = [], 0
            for keys in ex.map(fn, range(nwarm)):
                for k in keys:
                    d = vko.derive(k)
                    hs += [bytes(d.h160_c), bytes(d.h160_u)]
                    nkeys += 1
            out[name] = {"planted": planted, "warmup": {"inputs": nwarm, "keys": nkeys, "flags": 3, "digest": wl.hexd(wl.digest_of_h160s(hs))},
                         "first_input": (mn if name == "multibit" else es)[0].decode()}
            print(name, "done", flush=True)
    json.dump(out, open(os.path.join(HERE, "bench_c5.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
