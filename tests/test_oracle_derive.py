"""Pin the oracle's EC / KeyDeriver / Matcher restatement (oracle/secp256k1.c, derive.c, codecs.c).

Reference KATs: src/derive.rs:179-224, src/matcher.rs:117-149, src/analyze/key_parser.rs:275-287.
Independent pins: python big-int EC (tests/pyref.py), BIP-173's key-1 address.
"""
import random

from tests import pyref

CHBS = bytes.fromhex("c4bbcb1fbec99d65bf59d85c8cb62ee2db963f0fe106f483d9afa73bd4e39a8a")
N = pyref.N


def test_field_ops_vs_python(vko):
    rnd = random.Random(7)
    P = pyref.P
    vals = [0, 1, 2, P - 1, P - 2, 2**255, 2**256 - 1, 0x1000003D1, P + 5] + [rnd.getrandbits(256) for _ in range(200)]
    for _ in range(300):
        a, b = rnd.choice(vals), rnd.choice(vals)
        got = int.from_bytes(vko.fe_mul(a.to_bytes(32, "big"), b.to_bytes(32, "big")), "big")
        assert got == (a * b) % P
    for a in vals:
        if a % P:
            assert int.from_bytes(vko.fe_inv(a.to_bytes(32, "big")), "big") == pow(a, -1, P)


def test_normalize_key_edge_cases(vko):
    # src/derive.rs:79-93
    cases = {0: 1, 1: 1, N - 1: N - 1, N: 1, N + 1: 1, N + 2: 2, 2**256 - 1: (2**256 - 1) % N, 12345: 12345}
    for k, exp in cases.items():
        assert int.from_bytes(vko.normalize_key(k.to_bytes(32, "big")), "big") == exp
        assert pyref.normalize(k.to_bytes(32, "big")) == exp
        assert vko.scalar_valid(k.to_bytes(32, "big")) == (1 <= k < N)


def test_pubkey_slow_fast_python_agree(vko):
    rnd = random.Random(8)
    ks = [1, 2, 3, 255, 256, 257, 2**128, N - 1, N - 2, 2**255, int.from_bytes(CHBS, "big")]
    ks += [rnd.getrandbits(256) % (N - 1) + 1 for _ in range(40)]
    ks += [1 << (8 * i) for i in range(32)] + [0xFF << (8 * i) for i in range(31)]
    for k in ks:
        kb = k.to_bytes(32, "big")
        slow, fast, py = vko.pubkey(kb, slow=True), vko.pubkey(kb), pyref.pubkeys(k)
        assert slow == py and fast == py, hex(k)


def test_pubkey_batch_agrees_with_double_and_add(vko):
    """The third k*G path (16-bit windows + Montgomery batch inversion; generator of tests/golden/milksad_2p32.json and the
    lean CPU baseline) against plain double-and-add and python integers, including every single-window digit shape."""
    rnd = random.Random(9)
    ks = [1, 2, 3, 0xFFFF, 0x10000, 0x10001, 2**128, N - 1, N - 2, 2**255, int.from_bytes(CHBS, "big")]
    ks += [rnd.getrandbits(256) % (N - 1) + 1 for _ in range(60)]
    ks += [1 << (16 * i) for i in range(16)] + [0xFFFF << (16 * i) for i in range(15)] + [rnd.getrandbits(16 * i + 5) + 1 for i in range(16)]
    kbs = [k.to_bytes(32, "big") for k in ks]
    got = vko.pubkey_batch(kbs)
    for k, kb, g in zip(ks, kbs, got):
        assert g == vko.pubkey(kb, slow=True) == pyref.pubkeys(k), hex(k)
    assert vko.pubkey_batch(kbs[:1]) == got[:1] and vko.pubkey_batch([]) == []


def test_batched_sweep_equals_plain_sweep(vko):
    """VKO_F_BATCHED changes only how a chunk's public keys are computed: counters, digest and hits are the plain sweep's."""
    rnd = random.Random(10)
    for kind, fl in ((vko.T_MILKSAD, 1), (vko.T_SHA256, 3), (vko.T_LCG, 3), (vko.T_SHA256_CHAIN, 2), (vko.T_XORSHIFT, 1)):
        sp = vko.spec(kind)
        for lo in (0, 2**32 - 1500, 2**31 - 700, rnd.randrange(2**32)):
            d = vko.derive(vko.transform_apply(sp, lo + 5)[0][1])
            m = vko.Matcher([d.p2pkh_compressed.decode(), d.p2pkh_uncompressed.decode()])
            a = vko.sweep(sp, lo, lo + 1499, flags=fl, matcher=m, nthreads=2)
            b = vko.sweep(sp, lo, lo + 1499, flags=fl | vko.F_BATCHED, matcher=m, nthreads=3)
            assert a == b and a[0]["matches"] >= 1, (kind, lo)


def test_derive_reference_known_key(vko):
    d = vko.derive(CHBS)
    # src/derive.rs:190-199
    assert d.wif_uncompressed == b"5KJvsngHeMpm884wtkJNzQGaCErckhHJBGFsvd3VyK5qMZXj3hS"
    assert d.p2pkh_uncompressed == b"1JwSSubhmg6iPtRjtyqhUYYH7bZg3Lfy1T"
    assert d.wif_compressed[:1] in (b"K", b"L") and d.p2wpkh.startswith(b"bc1q")
    # src/analyze/key_parser.rs:283-287: this WIF decodes to the same key
    assert d.wif_compressed == b"L3p8oAcQTtuokSCRHQ7i4MhjWc9zornvpJLfmg62sYpLRJF9woSu"
    dec = vko.base58check_decode("L3p8oAcQTtuokSCRHQ7i4MhjWc9zornvpJLfmg62sYpLRJF9woSu")
    assert dec == b"\x80" + CHBS + b"\x01"
    assert vko.base58check_decode("5KJvsngHeMpm884wtkJNzQGaCErckhHJBGFsvd3VyK5qMZXj3hS") == b"\x80" + CHBS
    # survey appendix B anchors
    assert bytes(d.h160_c).hex() == "79fbfc3f34e7745860d76137da68f362380c606c"
    assert bytes(d.h160_u).hex() == "c4c5d791fcb4654a1ef5e03fe0ad3d9c598f9827"
    assert d.p2pkh_compressed == b"1C7zdTfnkzmr13HfA2vNm5SJYRK6nEKyq8"
    assert d.p2wpkh == b"bc1q08alc0e5ua69scxhvyma568nvguqccrv4cc9n4"
    assert d.pubkey_compressed == b"0378d430274f8c5ec1321338151e9f27f4c676a008bdf8638d07c0b6be9ab35c71"
    assert d.private_key_hex == CHBS.hex().encode()


def test_derive_zero_key_normalised(vko):
    # src/derive.rs:216-224
    d = vko.derive(bytes(32))
    assert d.private_key_decimal == b"1" and d.bit_length == 1
    assert d.hamming_weight == 1 and d.leading_zeros == 63
    # key 1 == BIP-173 example
    assert d.p2wpkh == b"bc1qw508d6qejxtdg4y5r3zarvary0c5xw7kv8f3t4"
    assert d.p2pkh_compressed == b"1BgGZ9tcN4rm9KBzDn7KprQz87SZ26SAMH"
    assert bytes(d.h160_c).hex() == "751e76e8199196d454941c45d1b3a323f1433bd6"


def test_derive_all_ones_key(vko):
    # src/derive.rs:204-212 and survey appendix B
    d = vko.derive(b"\x01" * 32)
    assert d.p2pkh_compressed == b"1C6Rc3w25VHud3dLDamutaqfKWqhrLRTaD"
    assert d.p2pkh_uncompressed == b"1BCwRkTsYzK5aNK4sdF7Bpti3PhrkPtLc4"
    assert d.p2wpkh == b"bc1q0xcqpzrky6eff2g52qdye53xkk9jxkvrh6yhyw"


def test_derive_record_vs_python(vko):
    rnd = random.Random(9)
    keys = [bytes(32), (N).to_bytes(32, "big"), (N + 1).to_bytes(32, "big"), b"\xff" * 32, (N - 1).to_bytes(32, "big")]
    keys += [rnd.getrandbits(256).to_bytes(32, "big") for _ in range(25)]
    keys += [rnd.getrandbits(rnd.randrange(1, 200)).to_bytes(32, "big") for _ in range(10)]
    for kb in keys:
        d = vko.derive(kb)
        k = pyref.normalize(kb)
        p33, p65 = pyref.pubkeys(k)
        assert bytes(d.raw) == k.to_bytes(32, "big")
        assert bytes(d.pub33) == p33 and bytes(d.pub65) == p65
        assert d.private_key_decimal == str(k).encode()
        assert d.private_key_binary == format(k, "0256b").encode()
        assert d.bit_length == k.bit_length() and d.hamming_weight == bin(k).count("1")
        hx = format(k, "064x")
        assert d.leading_zeros == len(hx) - len(hx.lstrip("0"))
        assert d.p2pkh_compressed.decode() == pyref.b58check(b"\0" + bytes(d.h160_c))
        assert d.p2pkh_uncompressed.decode() == pyref.b58check(b"\0" + bytes(d.h160_u))
        assert d.wif_compressed.decode() == pyref.b58check(b"\x80" + bytes(d.raw) + b"\x01")
        assert d.wif_uncompressed.decode() == pyref.b58check(b"\x80" + bytes(d.raw))
        assert vko.hash160(p33) == bytes(d.h160_c) and vko.hash160(p65) == bytes(d.h160_u)
        assert vko.segwit_decode("bc", d.p2wpkh.decode()) == (0, bytes(d.h160_c))
        assert vko.base58check_decode(d.p2pkh_compressed.decode()) == b"\0" + bytes(d.h160_c)
        assert vko.derive_h160(kb) == (bytes(d.h160_c), bytes(d.h160_u))


def test_codecs_reject_garbage(vko):
    assert vko.base58check_decode("1NonExistentAddress") is None
    assert vko.base58check_decode("1JwSSubhmg6iPtRjtyqhUYYH7bZg3Lfy1U") is None  # bad checksum
    assert vko.base58check_decode("0OIl") is None
    assert vko.segwit_decode("bc", "bc1qw508d6qejxtdg4y5r3zarvary0c5xw7kv8f3t5") is None
    assert vko.segwit_decode("bc", "BC1QW508D6QEJXTDG4Y5R3ZARVARY0C5XW7KV8F3T4") == (0, bytes.fromhex("751e76e8199196d454941c45d1b3a323f1433bd6"))
    assert vko.segwit_decode("tb", "bc1qw508d6qejxtdg4y5r3zarvary0c5xw7kv8f3t4") is None


def test_scalar_add(vko):
    rnd = random.Random(10)
    for _ in range(50):
        a, b = rnd.getrandbits(256), rnd.getrandbits(256)
        r = vko.scalar_add(a.to_bytes(32, "big"), b.to_bytes(32, "big"))
        exp = (a + b) % N
        assert (r is None and exp == 0) or int.from_bytes(r, "big") == exp
    assert vko.scalar_add((N - 5).to_bytes(32, "big"), (5).to_bytes(32, "big")) is None


def test_matcher_reference_cases(vko):
    # src/matcher.rs:117-138
    d = vko.derive(CHBS)
    m = vko.Matcher(["1JwSSubhmg6iPtRjtyqhUYYH7bZg3Lfy1T"])
    assert m.check(d) == 1
    assert m.check_h160(bytes(d.h160_c), bytes(d.h160_u)) == 1
    # src/matcher.rs:141-149
    m2 = vko.Matcher(["1NonExistentAddress"])
    assert m2.check(vko.derive(b"\x01" * 32)) == -1


def test_matcher_priority_and_load_rules(vko):
    d = vko.derive(CHBS)
    c, u, w = d.p2pkh_compressed.decode(), d.p2pkh_uncompressed.decode(), d.p2wpkh.decode()
    assert vko.Matcher([w, u, c]).check(d) == 0           # src/matcher.rs:74-79 first
    assert vko.Matcher([w, u]).check(d) == 1
    assert vko.Matcher([w]).check(d) == 2
    assert vko.Matcher([w.upper()]).check(d) == -1        # exact string match only
    # load(): trim, skip blank and '#', duplicates collapse (src/matcher.rs:50-56,101-103)
    m = vko.Matcher(["  " + c + " \r\n", "", "   ", "# " + u, c, "junk line"])
    assert m.count() == 2 and m.check(d) == 0
    many = vko.Matcher([f"addr{i}" for i in range(5000)] + [u])
    assert many.count() == 5001 and many.check(d) == 1


def test_sweep_small_vs_python(vko):
    # RangeSource::process semantics (src/source/range.rs:26-79): inclusive range, counters
    sp = vko.spec(vko.T_MILKSAD)
    st, hits = vko.sweep(sp, 10, 40, flags=vko.F_COMPRESSED, nthreads=3)
    assert (st["inputs"], st["keys"], st["matches"]) == (31, 31, 0) and hits == []
    dg = [0, 0, 0, 0]
    for s in range(10, 41):
        hc, _ = vko.derive_h160(vko.milksad_key(s))
        w = [int.from_bytes(hc[4 * i:4 * i + 4], "little") for i in range(5)]
        a, b = w[0] | w[1] << 32, w[2] | w[3] << 32
        dg[0] ^= a; dg[1] ^= b; dg[2] ^= w[4]; dg[3] = (dg[3] + a + b + w[4]) & (2**64 - 1)
    assert st["digest"] == tuple(dg)
    # planted targets of all three kinds; multi-key transform; thread-count independence
    sp = vko.spec(vko.T_SHA256)
    d1 = vko.derive(vko.transform_key(sp, 1003, 0))
    d2 = vko.derive(vko.transform_key(sp, 2999, 1))
    d3 = vko.derive(vko.transform_key(sp, 1000, 2))
    m = vko.Matcher([d1.p2pkh_compressed.decode(), d2.p2pkh_uncompressed.decode(), d3.p2wpkh.decode()])
    ref = None
    for nt in (1, 4):
        for flags in (vko.F_COMPRESSED | vko.F_UNCOMPRESSED, vko.F_COMPRESSED | vko.F_UNCOMPRESSED | vko.F_FAITHFUL):
            st, hits = vko.sweep(sp, 1000, 2999, flags=flags, matcher=m, nthreads=nt)
            assert (st["inputs"], st["keys"], st["matches"]) == (2000, 6000, 3)
            got = [(h["seed"], h["sub_index"], h["addr_kind"]) for h in hits]
            assert got == [(1000, 2, 2), (1003, 0, 0), (2999, 1, 1)]
            assert hits[1]["privkey"] == bytes(d1.raw) and hits[2]["h160"] == bytes(d2.h160_u)
            ref = ref or st
            assert st == ref
    # LCG all-variants key count at the minstd boundary (src/transform/lcg.rs:52)
    st, _ = vko.sweep(vko.spec(vko.T_LCG), 2**31 - 3, 2**31 + 1, nthreads=2)
    assert (st["inputs"], st["keys"]) == (5, 4 + 4 + 3 + 2 + 2)
