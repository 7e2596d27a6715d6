#!/usr/bin/env python3
"""Build-time lint for an LLVM register-coalescer miscompile that hit the generated field code (round 1, "short fold").

Root cause found in round 2 (profiles/archive/r02_short_fold_root_cause.txt): when outputs of two inline-asm statements are copied
into the halves of one wide register tuple (e.g. r.v[0..2] from the fold statement, r.v[3] from the carry chain before
it), the coalescer may join the 32-bit values into sub-registers of different tuples and then mark the asm output that fed
the dropped lane `dead` -- the lane of the surviving tuple keeps a stale value.  No hazard, no missing clobber: the MIR is
right before `register-coalescer` and wrong after it.

The signature is mechanical: an INLINEASM def that is live before the pass and `dead` after it.  This script compiles a
translation unit with -print-before/-print-after=register-coalescer and reports every such def.

    python tools/lint_coalescer.py [--group N ...]        # default: all sweep groups + the engine TU
Exit status 1 if anything is flagged.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vuke_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-Wno-unused-variable",
         "-mllvm", "-pragma-unroll-threshold=200000", "--cuda-device-only", "-S"]
DEF = re.compile(r"def (dead )?(undef )?(early-clobber )?(%\d+|\$\w+)")
HDR = re.compile(r"^# \*\*\* IR Dump (Before|After) Register Coalescer")
FN = re.compile(r"^# Machine code for function (\S+):")
SLOT = re.compile(r"^(\d+B)\s+INLINEASM ")
TOTAL = [0]


def parse(path):
    """{(function, when): {slot: [dead?, ...]}}"""
    out = {}
    when = fn = None
    with open(path, errors="replace") as f:
        for line in f:
            m = HDR.match(line)
            if m:
                when = m.group(1)
                continue
            m = FN.match(line)
            if m:
                fn = m.group(1)
                continue
            m = SLOT.match(line)
            if m and when and fn:
                ops = line.split("[attdialect]", 1)[-1]
                out.setdefault((fn, when), {})[m.group(1)] = [bool(d[0]) for d in DEF.findall(ops)]
    return out


def lint(src, extra, tag):
    with tempfile.TemporaryDirectory() as td:
        mir = os.path.join(td, "mir.txt")
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + extra + ["-o", os.path.join(td, "dev.s"), src,
               "-mllvm", "-print-before=register-coalescer", "-mllvm", "-print-after=register-coalescer"]
        with open(mir, "w") as f:
            subprocess.run(cmd, check=True, stderr=f, stdout=subprocess.DEVNULL)
        d = parse(mir)
    bad = 0
    nasm = 0
    unmatched = 0
    for (fn, when), slots in sorted(d.items()):
        if when != "Before":
            continue
        after = d.get((fn, "After"), {})
        for slot, flags in slots.items():
            nasm += 1
            fa = after.get(slot)
            if fa is None or len(fa) != len(flags):
                # the statement cannot be compared (gone after the pass, or its operand list changed): exactly where a dropped
                # output could hide, so it counts as a finding -- the shipped sources have none
                unmatched += 1
                print(f"{tag}: {fn} INLINEASM@{slot}: not comparable across register-coalescer ({'missing' if fa is None else 'operand count differs'})")
                continue
            for i, (b, a) in enumerate(zip(flags, fa)):
                if a and not b:
                    bad += 1
                    print(f"{tag}: {fn} INLINEASM@{slot}: output ${i} live before register-coalescer, dead after")
    print(f"{tag}: {nasm} asm statements checked, {bad} dropped outputs, {unmatched} not comparable")
    return bad, unmatched, nasm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--group", type=int, action="append")
    ap.add_argument("--no-engine", action="store_true")
    ap.add_argument("--csrc", default=CSRC)
    args = ap.parse_args()
    extra = os.environ.get("VK_EXTRA_FLAGS", "").split()   # the same variant flags build.sh takes
    jobs = [(os.path.join(args.csrc, "vk_sweep_tu.hip"), [f"-DVK_SWEEP_GROUP={g}"] + extra, f"sweep_g{g}") for g in (args.group if args.group is not None else [0, 1, 2])]
    if not args.no_engine and args.group is None:
        jobs.append((os.path.join(args.csrc, "vk_engine.hip"), extra, "engine"))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=int(os.environ.get("VK_LINT_JOBS", "3"))) as ex:   # each job is one hipcc process + ~2 GB of MIR text
        res = list(ex.map(lambda j: lint(*j), jobs))
    bad = sum(r[0] for r in res)
    unmatched = sum(r[1] for r in res)
    TOTAL[0] = sum(r[2] for r in res)
    if args.group is None and not args.no_engine and args.csrc == CSRC and not extra:
        # the record tests/test_abi.py checks against the kernel sources in the tree
        sys.path.insert(0, ROOT)
        import bench
        import json
        rec = {"tool": "tools/lint_coalescer.py", "csrc_hash": bench.csrc_hash(), "dropped_outputs": bad, "not_comparable": unmatched, "asm_statements": TOTAL[0],
               "translation_units": ["vk_sweep_tu.hip -DVK_SWEEP_GROUP=0..2", "vk_engine.hip"]}
        json.dump(rec, open(os.path.join(ROOT, "profiles", "lint_coalescer.json"), "w"), indent=1)
        print("wrote profiles/lint_coalescer.json")
    sys.exit(1 if (bad or unmatched) else 0)


if __name__ == "__main__":
    main()
