#!/usr/bin/env python3
"""tests/golden/labels.json: what Transform::apply_batch appends to the input's decimal string in the label of every slot, and what
Transform::name() returns -- for every numeric transform kind x variant x endian x chain depth, made by the oracle
(vko.transform_label / vko.transform_name, restating src/transform/lcg.rs:54-66, xorshift.rs:43-47, sha256_chain.rs:67,88).

The three places that must say the same are held to this file by tests/test_labels.py: the oracle itself, the C++ host
(vuke_amd/host/transform.cpp through vkh_label) and the Rust binding crate's `label_suffix` (bindings/rust/src/hip_range.rs),
which cannot be compiled here and is therefore replayed from its source text.

When /root/reference is present (the build container), the variant names are also compared with the strings of the reference's
own `name()` functions / `name:` fields, read from its source as text.

    python tests/golden/make_labels.py            # rewrites tests/golden/labels.json
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import vko  # noqa: E402

KINDS = [("sha256", vko.T_SHA256), ("double_sha256", vko.T_DOUBLE_SHA256), ("md5", vko.T_MD5), ("direct", vko.T_DIRECT), ("armory", vko.T_ARMORY),
         ("milksad", vko.T_MILKSAD), ("mt64", vko.T_MT64), ("lcg", vko.T_LCG), ("xorshift", vko.T_XORSHIFT), ("sha256_chain", vko.T_SHA256_CHAIN)]
SEEDS = (7, 1234567)        # the suffix must not depend on the input


def rows():
    out = []
    for kname, kind in KINDS:
        variants = (vko.VARIANT_ALL, 0, 1, 2, 3) if kind in (vko.T_LCG, vko.T_XORSHIFT, vko.T_SHA256_CHAIN) else (vko.VARIANT_ALL,)
        endians = (0, 1) if kind == vko.T_LCG else (0,)
        depths = (1, 3, 10) if kind == vko.T_SHA256_CHAIN else (10,)
        for v in variants:
            for en in endians:
                for d in depths:
                    sp = vko.spec(kind, v, en, d)
                    slots = vko.slots_per_input(sp)
                    suffixes = []
                    for sub in range(slots):
                        per_seed = []
                        for seed in SEEDS:
                            lab = vko.transform_label(sp, seed, sub)
                            assert lab is not None and lab.startswith(str(seed)), (kname, v, en, d, sub, lab)
                            per_seed.append(lab[len(str(seed)):])
                        assert len(set(per_seed)) == 1
                        suffixes.append(per_seed[0])
                    out.append({"kind": kname, "kind_id": kind, "variant": v, "endian": en, "chain_depth": d, "name": vko.transform_name(sp),
                                "slots": slots, "suffix": suffixes})
    return out


def reference_names():
    """Variant names as the reference spells them, read from its sources as text (None when the tree is absent)."""
    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        return None
    xs = re.findall(r'XorshiftVariant::\w+ => "([^"]+)"', open(os.path.join(ref, "xorshift.rs")).read().split("pub fn name")[1].split("}")[0])
    lcg = re.findall(r'^\s*name: "([^"]+)"', open(os.path.join(ref, "lcg.rs")).read(), re.M)
    ch = re.findall(r'=> "([^"]+)"', open(os.path.join(ref, "sha256_chain.rs")).read().split("pub fn name")[1].split("}\n")[0])
    return {"xorshift": xs, "lcg": lcg, "sha256_chain": ch}


def main():
    vko.build()
    data = {"made_by": "tests/golden/make_labels.py (oracle: vko.transform_label / transform_name)", "rows": rows()}
    names = reference_names()
    if names is not None:
        for r in data["rows"]:
            if r["variant"] == vko.VARIANT_ALL and r["kind"] in names and r["chain_depth"] in (1, 10) and r["endian"] == 0:
                if r["kind"] == "sha256_chain":
                    got = [s[1:].rsplit(":", 1)[0] for s in r["suffix"][:4 * r["chain_depth"]:r["chain_depth"]]]
                else:
                    got = [s[1:] for s in r["suffix"]]
                assert got == names[r["kind"]], (r["kind"], got, names[r["kind"]])
        data["reference_names_checked"] = names
    with open(os.path.join(ROOT, "tests", "golden", "labels.json"), "w") as f:
        json.dump(data, f, indent=1)
        f.write("\n")
    print("labels.json:", len(data["rows"]), "rows", "(reference names checked)" if names else "(reference tree absent)")


if __name__ == "__main__":
    main()
