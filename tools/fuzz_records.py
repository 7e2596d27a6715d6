#!/usr/bin/env python3
"""Randomised parity soak of the device's record formatter (vk_derive_columns, vk_generate_verbose, vk_generate_lines_strings) on the
GPU box: random keys of random magnitude (every bit length, runs of zero / 0xff bytes, values around n), every column against the
oracle's DerivedKey; verbose blocks and string-labelled lines of random inputs likewise.  usage: tools/fuzz_records.py [keys] [seed]"""
import os, random, struct, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi
from oracle import vko

total = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
vko.build()
eng = vuke_amd.Engine(0, window_bits=16)
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
t0 = time.time()
bad = 0
FIELDS = [("private_key_hex", "private_key_hex"), ("private_key_decimal", "private_key_decimal"), ("private_key_binary", "private_key_binary"),
          ("pubkey_compressed", "pubkey_compressed"), ("pubkey_uncompressed", "pubkey_uncompressed"), ("address_p2pkh_compressed", "p2pkh_compressed"),
          ("address_p2pkh_uncompressed", "p2pkh_uncompressed"), ("address_p2wpkh", "p2wpkh"), ("wif_compressed", "wif_compressed"),
          ("wif_uncompressed", "wif_uncompressed")]


def key():
    r = rnd.random()
    if r < 0.3:
        v = rnd.getrandbits(256)
    elif r < 0.6:
        v = rnd.getrandbits(rnd.randrange(1, 257))
    elif r < 0.7:
        v = (N + rnd.randrange(-5, 6)) % (1 << 256)
    elif r < 0.85:  # runs of zero / ff bytes
        b = bytearray(rnd.getrandbits(8) for _ in range(32))
        a, l = rnd.randrange(32), rnd.randrange(1, 33)
        b[a:a + l] = bytes([rnd.choice([0, 0xff])]) * len(b[a:a + l])
        v = int.from_bytes(b, "big")
    else:
        v = 1 << rnd.randrange(256)
    return v.to_bytes(32, "big")


def plain(vals):
    return b"".join(struct.pack("<I", len(v)) + v for v in vals)


done = 0
while done < total:
    n = min(rnd.choice([1, 63, 64, 65, 1000, 4096, 20000]), total - done)
    keys = [key() for _ in range(n)]
    got, rows = eng.derive_columns(keys)
    ds = [vko.derive(k) for k in keys]
    ok = rows == n and got["private_key_raw"] == b"".join(bytes(d.raw) for d in ds)
    for col, f in FIELDS:
        ok = ok and got[col] == plain([getattr(d, f) for d in ds])
    ok = ok and got["private_key_bit_length"] == b"".join(struct.pack("<I", d.bit_length) for d in ds)
    ok = ok and got["private_key_hamming_weight"] == b"".join(struct.pack("<I", d.hamming_weight) for d in ds)
    ok = ok and got["private_key_leading_zeros"] == b"".join(struct.pack("<I", d.leading_zeros) for d in ds)
    if not ok:
        bad += 1
        print("MISMATCH in a batch of", n, "keys; first key", keys[0].hex(), flush=True)
    done += n
print(f"vk_derive_columns: {done} keys, {bad} failing batches, {time.time() - t0:.0f} s", flush=True)

# verbose blocks and string-labelled lines of random inputs
vb = 0
for it in range(40):
    lo = rnd.getrandbits(rnd.randrange(1, 64))
    hi = lo + rnd.randrange(0, 300)
    text, st = eng.generate_lines(ffi.spec(ffi.T_SHA256), lo, hi, "sha256", [""] * 3, verbose=True)
    want = []
    for seed in range(lo, hi + 1):
        for sub in range(3):
            d = vko.derive(vko.transform_key(vko.spec(vko.T_SHA256), seed, sub))
            want.append(f"---\nsource: {seed}\ntransform: sha256\nprivate_key: {d.private_key_hex.decode()}\nwif_compressed: {d.wif_compressed.decode()}\n"
                        f"wif_uncompressed: {d.wif_uncompressed.decode()}\np2pkh_compressed: {d.p2pkh_compressed.decode()}\n"
                        f"p2pkh_uncompressed: {d.p2pkh_uncompressed.decode()}\np2wpkh: {d.p2wpkh.decode()}\n")
    if text.decode() != "".join(want):
        vb += 1
        print("VERBOSE MISMATCH", lo, hi, flush=True)
sb = 0
for it in range(40):
    strs = [bytes(rnd.randrange(1, 256) for _ in range(rnd.choice([1, 2, 31, 32, 33, 64, 100, 191, 192, 193, 400, 3000]))) for _ in range(rnd.randrange(1, 400))]
    kind, name = rnd.choice([(ffi.T_SHA256, "sha256"), (ffi.T_MD5, "md5"), (ffi.T_DIRECT, "direct"), (ffi.T_DOUBLE_SHA256, "double_sha256")])
    text, st = eng.generate_lines_strings(ffi.spec(kind), strs, name, [""])
    want = []
    for s in strs:
        k = vko.transform_string(vko.spec(kind), s, 0)
        if k is not None:
            d = vko.derive(k)
            want.append(s + b"," + name.encode() + b"," + d.private_key_hex + b"," + d.p2pkh_compressed + b"\n")
    if text != b"".join(want):
        sb += 1
        print("STRING LINES MISMATCH", kind, len(strs), flush=True)
print(f"vk_generate_verbose: 40 ranges, {vb} failing; vk_generate_lines_strings: 40 batches, {sb} failing; {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad or vb or sb else 0)
