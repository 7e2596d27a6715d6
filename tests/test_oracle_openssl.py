"""A third-party anchor for row a9 (KeyDeriver::derive) and for the committed goldens: OpenSSL's libcrypto (secp256k1 scalar
multiplication, SEC1 encodings, SHA-256, RIPEMD-160 from its legacy provider) against the oracle -- an implementation none of this
repository's authors wrote.  The reference reaches libsecp256k1 and bitcoin_hashes at the same boundary (src/derive.rs:97-118); the
other pins are python integers (tests/pyref.py), hashlib and published vectors.  Skipped where the image has no libcrypto with the
curve.  MT19937 (Milksad's generator, src/transform/milksad.rs:17-29) comes from numpy's RandomState here, so the planted golden of
BASELINE config 2 is re-derived end to end -- seed -> key -> public key -> hash160 -- without a line of this repository's arithmetic."""
import ctypes as C
import ctypes.util
import hashlib
import json
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


class OpenSSL:
    def __init__(self):
        path = ctypes.util.find_library("crypto")
        if not path:
            pytest.skip("no libcrypto in this image")
        L = self.L = C.CDLL(path)
        vp = C.c_void_p
        for name, res, args in (("EC_GROUP_new_by_curve_name", vp, [C.c_int]), ("EC_POINT_new", vp, [vp]), ("EC_POINT_free", None, [vp]),
                                ("BN_bin2bn", vp, [C.c_char_p, C.c_int, vp]), ("BN_free", None, [vp]), ("BN_CTX_new", vp, []),
                                ("EC_POINT_mul", C.c_int, [vp, vp, vp, vp, vp, vp]),
                                ("EC_POINT_point2oct", C.c_size_t, [vp, vp, C.c_int, C.c_char_p, C.c_size_t, vp]),
                                ("OSSL_PROVIDER_load", vp, [vp, C.c_char_p]), ("EVP_MD_fetch", vp, [vp, C.c_char_p, C.c_char_p]),
                                ("EVP_Digest", C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_uint), vp, vp])):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        self.group = L.EC_GROUP_new_by_curve_name(714)       # NID_secp256k1
        if not self.group:
            pytest.skip("this libcrypto has no secp256k1")
        self.ctx = L.BN_CTX_new()
        L.OSSL_PROVIDER_load(None, b"default")
        self.rmd = L.EVP_MD_fetch(None, b"RIPEMD160", None) if L.OSSL_PROVIDER_load(None, b"legacy") else None

    def pubkeys(self, k):
        """(33-byte, 65-byte) SEC1 encodings of k * G."""
        L = self.L
        bn = L.BN_bin2bn(k.to_bytes(32, "big"), 32, None)
        pt = L.EC_POINT_new(self.group)
        assert L.EC_POINT_mul(self.group, pt, bn, None, None, self.ctx) == 1
        out = []
        for form, size in ((2, 33), (4, 65)):
            buf = C.create_string_buffer(size)
            assert L.EC_POINT_point2oct(self.group, pt, form, buf, size, self.ctx) == size
            out.append(buf.raw)
        L.EC_POINT_free(pt)
        L.BN_free(bn)
        return out

    def hash160(self, data):
        if not self.rmd:
            pytest.skip("OpenSSL's legacy provider (RIPEMD-160) is not installed")
        out, n = C.create_string_buffer(20), C.c_uint(0)
        d = hashlib.sha256(data).digest()       # hashlib's SHA-256 is OpenSSL's
        assert self.L.EVP_Digest(d, 32, out, C.byref(n), self.rmd, None) == 1 and n.value == 20
        return out.raw


@pytest.fixture(scope="module")
def ossl():
    return OpenSSL()


def test_public_keys_and_hash160_against_openssl(vko, ossl):
    rnd = random.Random(2026)
    ks = [1, 2, 3, N - 1, N - 2, 2**255, 2**128, (1 << 256) // 3 % N, int("c4bbcb1fbec99d65bf59d85c8cb62ee2db963f0fe106f483d9afa73bd4e39a8a", 16)]
    ks += [1 << i for i in range(0, 256, 7)] + [(1 << i) - 1 for i in range(8, 256, 9)]
    ks += [rnd.getrandbits(256) % (N - 1) + 1 for _ in range(3000)]
    for k in ks:
        kb = k.to_bytes(32, "big")
        p33, p65 = ossl.pubkeys(k)
        assert vko.pubkey(kb) == (p33, p65), hex(k)
        assert vko.derive_h160(kb) == (ossl.hash160(p33), ossl.hash160(p65)), hex(k)
    # keys outside [1, n): KeyDeriver::derive reduces them first (src/derive.rs:79-93)
    for raw in (0, N, N + 1, 2**256 - 1):
        k = raw % N or 1
        p33, p65 = ossl.pubkeys(k)
        assert vko.derive_h160(raw.to_bytes(32, "big")) == (ossl.hash160(p33), ossl.hash160(p65))


def test_planted_milksad_golden_against_numpy_and_openssl(ossl):
    """tests/golden/milksad_planted.json (the 1026 planted seeds of BASELINE config 2, which every bench run and the full-size GPU tests
    match hits against): key = the first 32 bytes of numpy's MT19937 stream, hash160 through OpenSSL."""
    np = pytest.importorskip("numpy")
    planted = json.load(open(os.path.join(ROOT, "tests", "golden", "milksad_planted.json")))["planted"]
    assert len(planted) == 1026
    for row in planted:
        key = np.random.RandomState(row["seed"]).bytes(32)      # init_genrand(seed), tempered outputs as little-endian words
        assert key.hex() == row["privkey"], row["seed"]
        k = int.from_bytes(key, "big")
        k = k % N or 1 if not 1 <= k < N else k
        assert ossl.hash160(ossl.pubkeys(k)[0]).hex() == row["h160_c"], row["seed"]


def test_oracle_sweep_window_against_numpy_and_openssl(vko, ossl):
    """The digest the GPU parity tests compare (XOR / sum fold of all hash160s of a range) recomputed for a Milksad window from numpy's
    MT19937 and OpenSSL's curve and hashes alone."""
    np = pytest.importorskip("numpy")
    lo, hi = 3_000_000_000, 3_000_000_000 + 1499
    d = [0, 0, 0, 0]
    for seed in range(lo, hi + 1):
        k = int.from_bytes(np.random.RandomState(seed).bytes(32), "big")
        k = k % N or 1 if not 1 <= k < N else k
        p33, p65 = ossl.pubkeys(k)
        for h in (ossl.hash160(p33), ossl.hash160(p65)):
            w = [int.from_bytes(h[4 * i:4 * i + 4], "little") for i in range(5)]
            a, b = w[0] | w[1] << 32, w[2] | w[3] << 32
            d[0] ^= a
            d[1] ^= b
            d[2] ^= w[4]
            d[3] = (d[3] + a + b + w[4]) % 2**64
    st, _ = vko.sweep(vko.spec(vko.T_MILKSAD), lo, hi, flags=3, nthreads=4)
    assert (st["keys"], list(st["digest"])) == (1500, d)
