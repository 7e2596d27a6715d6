#!/usr/bin/env python3
"""Generate tests/golden/c1_generate_sha256_0_1000000.json: BASELINE config 1 at its stated size through the oracle.

`vuke generate --transform=sha256 range --start 0 --end 1000000` prints 3 000 003 compact lines
`{label},{transform},{privkey_hex},{p2pkh_compressed}` (src/output/console.rs:77-81) in a thread-dependent order
(src/source/range.rs:42), so the fixture is order-free: the line count, the SHA-256 of the sorted output, and the
first / last sorted lines.  Keys: vko.transform_key; public keys: vko_pubkey_batch (held to the double-and-add path
in tests/test_oracle_derive.py); hash160 and Base58Check: the oracle's own.  A few minutes of CPU.

    python tests/golden/make_c1.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import vko  # noqa: E402

LO, HI = 0, 1_000_000


def main():
    sp = vko.spec(vko.T_SHA256)
    lines = []
    chunk = 20000
    for base in range(LO, HI + 1, chunk):
        items = []
        for s in range(base, min(base + chunk, HI + 1)):
            for sub in range(3):
                k = vko.transform_key(sp, s, sub)
                if k is not None:
                    items.append((str(s), vko.normalize_key(k)))
        pubs = vko.pubkey_batch([k for _, k in items])
        for (label, k), (p33, _) in zip(items, pubs):
            addr = vko.base58check_encode(b"\0" + vko.hash160(p33))
            lines.append(f"{label},sha256,{k.hex()},{addr}")
        print(base, flush=True)
    # spot-check the batched lines against the full KeyDeriver restatement
    for i in (0, 1, 2, len(lines) // 2, len(lines) - 1):
        label, _, kh, addr = lines[i].split(",")
        d = vko.derive(bytes.fromhex(kh))
        assert d.private_key_hex.decode() == kh and d.p2pkh_compressed.decode() == addr
    lines.sort()
    blob = ("\n".join(lines) + "\n").encode()
    out = {
        "what": "config 1: generate --transform=sha256 range 0..=1000000, compact lines, sorted",
        "generator": "tests/golden/make_c1.py",
        "inputs": HI - LO + 1, "lines": len(lines), "sorted_sha256": hashlib.sha256(blob).hexdigest(), "bytes": len(blob),
        "first": lines[0], "last": lines[-1],
    }
    json.dump(out, open(os.path.join(HERE, "c1_generate_sha256_0_1000000.json"), "w"), indent=0)
    print(out)


if __name__ == "__main__":
    main()
