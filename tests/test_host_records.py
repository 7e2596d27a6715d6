"""The device's text / record formatters (vk_lines.cuh, vk_records.cuh) on a machine without a GPU: the slot-writing kernels compiled
for the host one lane per item (tests/host_emu/host_records_emu.cpp) under AddressSanitizer + UBSan, held to python's own encodings:
Base58Check from hashlib and int arithmetic, BIP-173 bech32 restated here, str / format of python integers.  The inputs a key's
record needs beside the key (hash160s, public key) come from the oracle.  What the GPU tests add is the wave-level compaction."""
import hashlib
import os
import random
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs ROCm's clang++ for the host build")
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
B58 = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz"
B32 = "qpzry9x8gf2tvdw0s3jn54khce6mua7l"


def base58check(payload):
    raw = payload + hashlib.sha256(hashlib.sha256(payload).digest()).digest()[:4]
    v, out = int.from_bytes(raw, "big"), ""
    while v:
        v, r = divmod(v, 58)
        out = B58[r] + out
    return "1" * (len(raw) - len(raw.lstrip(b"\0"))) + out


def bech32_p2wpkh(h160):
    def polymod(values):
        gen, chk = [0x3b6a57b2, 0x26508e6d, 0x1ea119fa, 0x3d4233dd, 0x2a1462b3], 1
        for v in values:
            b = chk >> 25
            chk = ((chk & 0x1ffffff) << 5) ^ v
            for i in range(5):
                chk ^= gen[i] if (b >> i) & 1 else 0
        return chk
    bits = int.from_bytes(h160, "big")
    data = [0] + [(bits >> (5 * (31 - i))) & 31 for i in range(32)]
    hrp = [ord(c) >> 5 for c in "bc"] + [0] + [ord(c) & 31 for c in "bc"]
    pm = polymod(hrp + data + [0] * 6) ^ 1
    return "bc1" + "".join(B32[d] for d in data + [(pm >> (5 * (5 - i))) & 31 for i in range(6)])


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    exe = tmp_path_factory.mktemp("rec_emu") / "rec_emu"
    subprocess.run([CLANG, "-std=c++17", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wno-unknown-pragmas", "-Wno-unknown-attributes",
                    "-I", os.path.join(ROOT, "vuke_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "host_emu"),
                    "-o", str(exe), os.path.join(ROOT, "tests", "host_emu", "host_records_emu.cpp")], check=True, capture_output=True, text=True)

    def run(lo, name, suffixes, items):
        """items: [(input, slot, key32, h160c, h160u, xy64, valid)] -> {(input, slot): {"lines", "verbose", 0..14}}"""
        text = "".join(f"{i} {s} {k.hex()} {c.hex()} {u.hex()} {p.hex()} {v}\n" for i, s, k, c, u, p, v in items)
        r = subprocess.run([str(exe), str(lo), str(len(suffixes)), name] + [x or "-" for x in suffixes], input=text, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-3000:])
        out, cur = {}, None
        for line in r.stdout.splitlines():
            f = line.split(" ")
            if f[0] == "item":
                cur = out.setdefault((int(f[1]), int(f[2])), {"valid": int(f[3])})
            elif f[0] == "col":
                cur[int(f[1])] = bytes.fromhex(f[2]) if len(f) > 2 else b""
            else:
                cur[f[0]] = bytes.fromhex(f[1]) if len(f) > 1 else b""
        return out
    return run


def want_record(source, name, d_raw, h160c, h160u, xy):
    k = int.from_bytes(d_raw, "big")
    plain = lambda s: struct.pack("<I", len(s)) + s.encode()
    wif_c, wif_u = base58check(b"\x80" + d_raw + b"\x01"), base58check(b"\x80" + d_raw)
    p2c, p2u, p2w = base58check(b"\0" + h160c), base58check(b"\0" + h160u), bech32_p2wpkh(h160c)
    pubc = ("03" if xy[63] & 1 else "02") + xy[:32].hex()
    hx = d_raw.hex()
    cols = {0: plain(source), 1: d_raw, 2: plain(hx), 3: plain(str(k)), 4: plain(format(k, "0256b")), 5: struct.pack("<I", k.bit_length()),
            6: struct.pack("<I", bin(k).count("1")), 7: struct.pack("<I", len(hx) - len(hx.lstrip("0"))), 8: plain(pubc), 9: plain("04" + xy.hex()),
            10: plain(p2c), 11: plain(p2u), 12: plain(p2w), 13: plain(wif_c), 14: plain(wif_u)}
    cols["lines"] = f"{source},{name},{hx},{p2c}\n".encode()
    cols["verbose"] = (f"---\nsource: {source}\ntransform: {name}\nprivate_key: {hx}\nwif_compressed: {wif_c}\nwif_uncompressed: {wif_u}\n"
                       f"p2pkh_compressed: {p2c}\np2pkh_uncompressed: {p2u}\np2wpkh: {p2w}\n").encode()
    return cols


def test_formatters_against_python_encodings(emu):
    from oracle import vko
    rnd = random.Random(4242)
    keys = [1, 2, 255, 256, N - 1, N - 2, 1 << 255, (1 << 252) - 1, 10**77, 10**77 - 1, int("f" * 63, 16), int("0" * 32 + "1" * 32, 16)]
    keys += [1 << b for b in (4, 8, 31, 32, 33, 63, 64, 128, 200, 248)]
    keys += [rnd.getrandbits(rnd.randrange(1, 257)) % N or 1 for _ in range(200)]
    suffixes = ["", ":glibc", "[indexed:be:1234567]", "s" * 31]
    name = "sha256_chain[indexed:be]"
    lo = 2**64 - 1 - (len(keys) + 3) // 4
    items, want = [], {}
    for j, k in enumerate(keys):
        d = vko.derive(k.to_bytes(32, "big"))
        raw, hc, hu, xy = bytes(d.raw), bytes(d.h160_c), bytes(d.h160_u), bytes(d.pub65)[1:]
        if j % 37 == 5:     # hash160s with leading zero bytes: one '1' per zero byte in the address
            hc, hu = bytes(j % 3 + 1) + hc[j % 3 + 1:], bytes(2) + hu[2:]
        if j == 7:
            hc = bytes(20)
        inp, slot = lo + j // 4, j % 4
        items.append((inp, slot, raw, hc, hu, xy, 1))
        want[(inp, slot)] = want_record(f"{inp}{suffixes[slot]}", name, raw, hc, hu, xy)
    items.append((lo + len(keys) // 4 + 1, 2, bytes(32), bytes(20), bytes(20), bytes(64), 0))    # an item without a key: nothing written
    got = emu(lo, name, suffixes, items)
    for key, w in want.items():
        g = got[key]
        for col, val in w.items():
            assert g[col] == val, (key, col)
    hole = got[(lo + len(keys) // 4 + 1, 2)]
    assert hole["valid"] == 0 and all(hole[c] == b"" for c in list(range(15)) + ["lines", "verbose"])
    # pinned vector of the reference's own tests (src/derive.rs:190-197): c4bb...9a8a -> 5KJvsng...3hS, 1JwSSub...y1T
    k = bytes.fromhex("c4bbcb1fbec99d65bf59d85c8cb62ee2db963f0fe106f483d9afa73bd4e39a8a")
    d = vko.derive(k)
    g = emu(0, "sha256", [""], [(0, 0, k, bytes(d.h160_c), bytes(d.h160_u), bytes(d.pub65)[1:], 1)])[(0, 0)]
    assert g[14][4:] == b"5KJvsngHeMpm884wtkJNzQGaCErckhHJBGFsvd3VyK5qMZXj3hS" and g[11][4:] == b"1JwSSubhmg6iPtRjtyqhUYYH7bZg3Lfy1T"
    assert g[12][4:] == d.p2wpkh and g[13][4:] == d.wif_compressed


def test_harness_shares_the_shipped_sources():
    for f in os.listdir(os.path.join(ROOT, "vuke_amd", "csrc")):
        if f.endswith((".cuh", ".hip", ".h")):
            assert "host_records_emu" not in open(os.path.join(ROOT, "vuke_amd", "csrc", f)).read()
