#!/usr/bin/env python3
"""Generate tests/golden/bench_c3_full.json and bench_c4_full.json: the CPU oracle over BASELINE configs 3 and 4 at their
stated sizes, block by block as bench.py steps over them.

  config 3  sha256 transform (3 keys per input, both hash160s), inputs 0 .. 3*2^25-1 (covers the config's 0..99 999 999),
            three blocks of 2^25 inputs: counters + digest per block
  config 4  all four LCG variants (BE), compressed hash160, the whole 2^31 seed space in 32 blocks of 2^26 seeds:
            counters + digest per block (minstd skips seed 2^31-1: 4*2^31 - 1 keys in all)
About 9 x 10^9 keys: half an hour on 16 fast cores.  Checkpointed per block in *.partial.jsonl (VK_PARTIAL_DIR).

    python tests/golden/make_c34_full.py [--threads 16] [--only 3|4]
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import vko  # noqa: E402
import bench_workloads as wl  # noqa: E402

PDIR = os.environ.get("VK_PARTIAL_DIR", HERE)


def run(tag, spec, flags, block_inputs, nblocks, threads, what):
    part = os.path.join(PDIR, f"bench_c{tag}_full.partial.jsonl")
    done = {}
    if os.path.exists(part):
        for line in open(part):
            r = json.loads(line)
            done[r["block"]] = r
    for b in range(nblocks):
        if b in done:
            continue
        lo = b * block_inputs
        t = time.time()
        st, _ = vko.sweep(spec, lo, lo + block_inputs - 1, flags=flags | vko.F_BATCHED, nthreads=threads)
        r = {"block": b, "lo": lo, "hi": lo + block_inputs - 1, "inputs": st["inputs"], "keys": st["keys"], "digest": wl.hexd(st["digest"]),
             "seconds": round(time.time() - t, 1)}
        with open(part, "a") as f:
            f.write(json.dumps(r) + "\n")
        done[b] = r
        print(tag, "block", b, r["seconds"], "s", flush=True)
    if len(done) == nblocks:
        blocks = [done[b] for b in range(nblocks)]
        out = {"what": what, "generator": "tests/golden/make_c34_full.py", "flags": flags, "block_inputs": block_inputs,
               "blocks": [{k: r[k] for k in ("lo", "hi", "inputs", "keys", "digest")} for r in blocks],
               "keys": sum(r["keys"] for r in blocks),
               "digest": wl.hexd(wl.fold_digests([wl.unhexd(r["digest"]) for r in blocks]))}
        json.dump(out, open(os.path.join(HERE, f"bench_c{tag}_full.json"), "w"), indent=0)
        print("wrote", f"bench_c{tag}_full.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    if args.only in ("", "3"):
        run("3", vko.spec(vko.T_SHA256), 3, wl.C3_STEP_INPUTS, 3, args.threads,
            "config 3 at size: sha256 transform, both hash160s, three blocks of 2^25 inputs (0 .. 100 663 295)")
    if args.only in ("", "4"):
        run("4", vko.spec(vko.T_LCG), 1, wl.C4_STEP_SEEDS, wl.C4_SEEDS // wl.C4_STEP_SEEDS, args.threads,
            "config 4 at size: all LCG variants x the whole 2^31 seed space, compressed hash160, 32 blocks of 2^26 seeds")


if __name__ == "__main__":
    main()
