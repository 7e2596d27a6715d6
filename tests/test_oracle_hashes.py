"""Pin the oracle's hash primitives (oracle/hashes.c) against hashlib and published vectors."""
import hashlib
import hmac
import random


def test_sha256_vs_hashlib(vko):
    rnd = random.Random(1)
    for n in list(range(0, 130)) + [255, 256, 1000]:
        m = bytes(rnd.getrandbits(8) for _ in range(n))
        assert vko.sha256(m) == hashlib.sha256(m).digest(), n


def test_double_sha256_empty_reference_kat(vko):
    # reference src/electrum.rs:239-245
    assert vko.sha256(vko.sha256(b"")).hex() == "5df6e0e2761359d30a8275058e299fcc0381534545f55cf43e41983f5d4c9456"


def test_sha512_hmac_pbkdf2_vs_hashlib(vko):
    rnd = random.Random(2)
    for n in list(range(0, 260, 7)) + [111, 112, 113, 127, 128, 129, 239, 240, 241]:
        m = bytes(rnd.getrandbits(8) for _ in range(n))
        assert vko.sha512(m) == hashlib.sha512(m).digest(), n
    for kl in (0, 1, 64, 127, 128, 129, 300):
        k = bytes(rnd.getrandbits(8) for _ in range(kl))
        m = bytes(rnd.getrandbits(8) for _ in range(kl + 3))
        assert vko.hmac_sha512(k, m) == hmac.new(k, m, hashlib.sha512).digest(), kl
    for pw, salt, it in ((b"password", b"salt", 1), (b"password", b"mnemonic", 2048), (b"x" * 300, b"mnemonicTREZOR", 17)):
        assert vko.pbkdf2_hmac_sha512(pw, salt, it) == hashlib.pbkdf2_hmac("sha512", pw, salt, it)


# RIPEMD-160: hashlib in this image has no ripemd160 (OpenSSL 3 legacy provider absent),
# so pin on the vectors published with the algorithm (Dobbertin, Bosselaers, Preneel 1996).
RMD_VECTORS = [
    (b"", "9c1185a5c5e9fc54612808977ee8f548b2258d31"),
    (b"a", "0bdc9d2d256b3ee9daae347be6f4dc835a467ffe"),
    (b"abc", "8eb208f7e05d987a9b044a8e98c6b087f15a0bfc"),
    (b"message digest", "5d0689ef49d2fae572b881b123a85ffa21595f36"),
    (b"abcdefghijklmnopqrstuvwxyz", "f71c27109c692c1b56bbdceb5b9d2865b3708dbc"),
    (b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq", "12a053384a9c0c88e405a06c27dcf49ada62eb2b"),
    (b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789", "b0e20b6e3116640286ed3a87a5713079b21f5189"),
    (b"1234567890" * 8, "9b752e45573d4b39f4dbd3323cab82bf63326bfb"),
    (b"a" * 1000000, "52783243c1697bdbe16d37f97f68f08325dc1528"),
]


def test_ripemd160_published_vectors(vko):
    for m, h in RMD_VECTORS:
        assert vko.ripemd160(m).hex() == h


def test_hash160_key1_bip173(vko):
    # compressed pubkey of private key 1 -> BIP-173's example program
    pub = bytes.fromhex("0279be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798")
    assert vko.hash160(pub).hex() == "751e76e8199196d454941c45d1b3a323f1433bd6"
