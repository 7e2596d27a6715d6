#!/usr/bin/env python3
"""Regenerate the golden fixtures in tests/golden/ from the CPU oracle.

The reference (Rust) cannot be built or run in this pipeline, so these vectors come from the
oracle AFTER it has passed every reference KAT (tests/test_oracle_*.py); see DESIGN.md "Oracle".
Fixtures are data only: inputs and expected outputs.

    python tests/golden/make_golden.py
"""
import json
import os
import random
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import vko  # noqa: E402

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
SEEDS = [0, 1, 2, 42, 100, 12345, 2**31 - 2, 2**31 - 1, 2**31, 2**32 - 1, 2**32, 2**64 - 1]


def derive_record(k):
    d = vko.derive(k)
    return {
        "key": k.hex(), "raw": bytes(d.raw).hex(), "pub33": bytes(d.pub33).hex(), "pub65": bytes(d.pub65).hex(),
        "h160_c": bytes(d.h160_c).hex(), "h160_u": bytes(d.h160_u).hex(),
        "p2pkh_compressed": d.p2pkh_compressed.decode(), "p2pkh_uncompressed": d.p2pkh_uncompressed.decode(),
        "p2wpkh": d.p2wpkh.decode(), "wif_compressed": d.wif_compressed.decode(), "wif_uncompressed": d.wif_uncompressed.decode(),
        "decimal": d.private_key_decimal.decode(), "bit_length": d.bit_length, "hamming_weight": d.hamming_weight,
        "leading_zeros": d.leading_zeros,
    }


def main():
    rnd = random.Random(20261003)
    keys = [bytes(32), (1).to_bytes(32, "big"), (N - 1).to_bytes(32, "big"), N.to_bytes(32, "big"), (N + 1).to_bytes(32, "big"),
            b"\xff" * 32, bytes.fromhex("c4bbcb1fbec99d65bf59d85c8cb62ee2db963f0fe106f483d9afa73bd4e39a8a"), b"\x01" * 32]
    keys += [rnd.getrandbits(256).to_bytes(32, "big") for _ in range(48)]
    keys += [rnd.getrandbits(rnd.randrange(1, 255)).to_bytes(32, "big") for _ in range(8)]
    json.dump([derive_record(k) for k in keys], open(os.path.join(HERE, "derive.json"), "w"), indent=0)

    specs = {
        "sha256": (vko.T_SHA256, vko.VARIANT_ALL, 0, 10), "milksad": (vko.T_MILKSAD, vko.VARIANT_ALL, 0, 10),
        "mt64": (vko.T_MT64, vko.VARIANT_ALL, 0, 10), "lcg": (vko.T_LCG, vko.VARIANT_ALL, 0, 10),
        "lcg:le": (vko.T_LCG, vko.VARIANT_ALL, 1, 10), "lcg:minstd": (vko.T_LCG, 1, 0, 10), "lcg:msvc:le": (vko.T_LCG, 2, 1, 10),
        "xorshift": (vko.T_XORSHIFT, vko.VARIANT_ALL, 0, 10), "xorshift:128": (vko.T_XORSHIFT, 1, 0, 10),
        "sha256_chain": (vko.T_SHA256_CHAIN, vko.VARIANT_ALL, 0, 10), "sha256_chain:counter": (vko.T_SHA256_CHAIN, 3, 0, 10),
        "sha256_chain:indexed:le": (vko.T_SHA256_CHAIN, 2, 0, 10),
        "double_sha256": (vko.T_DOUBLE_SHA256, vko.VARIANT_ALL, 0, 10), "md5": (vko.T_MD5, vko.VARIANT_ALL, 0, 10),
        "direct": (vko.T_DIRECT, vko.VARIANT_ALL, 0, 10), "armory": (vko.T_ARMORY, vko.VARIANT_ALL, 0, 10),
    }
    tv = {}
    for name, (kind, var, en, depth) in specs.items():
        sp = vko.spec(kind, var, en, depth)
        tv[name] = {"spec": [kind, var, en, depth], "name": vko.transform_name(sp),
                    "vectors": {str(s): [[l, k.hex()] for l, k in vko.transform_apply(sp, s)] for s in SEEDS}}
    json.dump(tv, open(os.path.join(HERE, "transforms.json"), "w"), indent=0)

    # passphrase inputs (Input::from_string): per transform, the keys each string emits (null = nothing emitted)
    srnd = random.Random(424242)
    strings = [b"", b"a", b"password", b"correct horse battery staple", "za\u017c\u00f3\u0142\u0107".encode(), b"x" * 32, b"y" * 33, b"z" * 55, b"w" * 56, b"v" * 64, b"u" * 119, b"t" * 120]
    strings += [bytes(srnd.randrange(1, 256) for _ in range(srnd.randrange(1, 300))) for _ in range(20)]
    sspecs = {"sha256": (vko.T_SHA256, vko.VARIANT_ALL, 0, 10), "double_sha256": (vko.T_DOUBLE_SHA256, vko.VARIANT_ALL, 0, 10),
              "md5": (vko.T_MD5, vko.VARIANT_ALL, 0, 10), "direct": (vko.T_DIRECT, vko.VARIANT_ALL, 0, 10),
              "armory": (vko.T_ARMORY, vko.VARIANT_ALL, 0, 10), "sha256_chain": (vko.T_SHA256_CHAIN, vko.VARIANT_ALL, 0, 3),
              "sha256_chain:indexed:be": (vko.T_SHA256_CHAIN, 1, 0, 5)}
    sv = {"strings": [x.hex() for x in strings], "transforms": {}}
    for name, (kind, var, en, depth) in sspecs.items():
        sp = vko.spec(kind, var, en, depth)
        rows = []
        for x in strings:
            ks = [vko.transform_string(sp, x, sub) for sub in range(vko.slots_per_string(sp))]
            rows.append([k.hex() if k is not None else None for k in ks])
        sv["transforms"][name] = {"spec": [kind, var, en, depth], "keys": rows}
    json.dump(sv, open(os.path.join(HERE, "strings.json"), "w"), indent=0)

    # config 1 in miniature: `generate --transform sha256 range --start 0 --end 1000`, compact lines, sorted
    sp = vko.spec(vko.T_SHA256)
    lines = []
    for s in range(0, 1001):
        for label, k in vko.transform_apply(sp, s):
            d = vko.derive(k)
            lines.append(f"{label},sha256,{d.private_key_hex.decode()},{d.p2pkh_compressed.decode()}")
    open(os.path.join(HERE, "e2e_generate_sha256_0_1000.csv"), "w").write("\n".join(sorted(lines)) + "\n")

    # scan_small: 1000 seeds, planted targets of mixed kinds + decoys, expected hits
    sp = vko.spec(vko.T_LCG)
    lo, hi = 20000, 20999
    targets, expect = [], []
    for j in range(45):
        seed, sub, kind = rnd.randrange(lo, hi + 1), rnd.randrange(4), j % 3
        d = vko.derive(vko.transform_key(sp, seed, sub))
        targets.append([d.p2pkh_compressed, d.p2pkh_uncompressed, d.p2wpkh][kind].decode())
    for j in range(200):
        h = vko.sha256(b"decoy" + struct.pack("<Q", j))[:20]
        targets.append(vko.base58check_encode(b"\0" + h) if j % 2 == 0 else vko.segwit_encode("bc", 0, h))
    targets += ["# comment", "", "not an address", "  " + targets[0] + "  "]
    m = vko.Matcher(targets)
    st, hits = vko.sweep(sp, lo, hi, flags=3, matcher=m, nthreads=4)
    json.dump({"transform": "lcg", "lo": lo, "hi": hi, "targets": targets, "target_count": m.count(),
               "stats": {"inputs": st["inputs"], "keys": st["keys"], "matches": st["matches"]},
               "digest": [f"{x:016x}" for x in st["digest"]],
               "hits": [{"seed": h["seed"], "sub_index": h["sub_index"], "addr_kind": h["addr_kind"], "privkey": h["privkey"].hex(),
                         "h160": h["h160"].hex(), "label": vko.transform_label(sp, h["seed"], h["sub_index"])} for h in hits]},
              open(os.path.join(HERE, "scan_small.json"), "w"), indent=0)

    # digest windows for parity at scale (2^20-seed Milksad windows, compressed only)
    wins = []
    for lo in (0, 123456789, 2**32 - 2**20):
        st, _ = vko.sweep(vko.spec(vko.T_MILKSAD), lo, lo + 2**20 - 1, flags=1, nthreads=os.cpu_count() or 1)
        wins.append({"lo": lo, "hi": lo + 2**20 - 1, "keys": st["keys"], "digest": [f"{x:016x}" for x in st["digest"]]})
    json.dump(wins, open(os.path.join(HERE, "digest_windows.json"), "w"), indent=0)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
