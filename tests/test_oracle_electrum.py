"""Pin the oracle's Electrum v1 restatement against the reference's vectors (src/electrum.rs:216-285)
and an independent hashlib transcription."""
import hashlib

from tests import pyref

SEED = b"acb740e454c3134901d7c8f16497cc1c"
MPK = "e9d4b7866dd1e91c862aebf62a49548c7dbf7bcc6e4b7b8c9da820c7737968df9c09d5a3e271dc814a29981f81b3faaf2737b551ef5dcc6189cf0f8252c442b3"


def py_stretch(seed, rounds):
    x = hashlib.sha256(seed + seed).digest()
    for _ in range(1, rounds):
        x = hashlib.sha256(x + seed).digest()
    return x


def test_stretch_vs_hashlib(vko):
    # any length: the reference hashes the whole slice (src/electrum.rs:150-168); 129 .. 1000 bytes were cut at 128 until round 3
    for seed in (b"", b"7", b"12345678901234567890", SEED, b"ab" * 32, b"f" * 23, b"f" * 24, b"c" * 128, b"d" * 129, b"e" * 300, b"9" * 1000):
        for rounds in (1, 2, 1000):
            assert vko.electrum_stretch(seed, rounds) == py_stretch(seed, rounds), (seed, rounds)


def test_reference_vectors(vko):
    mpk, keys = vko.electrum_keys(SEED)
    assert mpk.hex() == MPK                                   # src/electrum.rs:258-264
    assert len(keys) == 20 and keys[0][0] == "acb740e4...6497cc1c[0/0]" and keys[19][0].endswith("[0/19]")
    d = vko.derive(keys[0][1])
    assert d.p2pkh_uncompressed == b"1FJEEB8ihPMbzs2SkLmr37dHyRFzakqUmo"   # :267-273
    _, ckeys = vko.electrum_keys(SEED, for_change=True)
    assert vko.derive(ckeys[0][1]).p2pkh_uncompressed == b"1KRW8pH6HFHZh889VDq6fEKvmrsmApwNfe"  # :276-284
    assert ckeys[3][0] == "acb740e4...6497cc1c[1/3]"
    assert vko.electrum_keys(b"not_valid_hex!") == (None, [])  # :299-303
    assert vko.electrum_keys(b"abcd1234", count=2, rounds=10)[1][0][0] == "abcd1234[0/0]"  # truncate_seed :305-312


def test_child_keys_vs_python(vko):
    master = int.from_bytes(py_stretch(SEED, 100000), "big")
    mpk = bytes.fromhex(MPK)
    _, keys = vko.electrum_keys(SEED, count=12)
    for i, (_, k) in enumerate(keys):
        seq = hashlib.sha256(hashlib.sha256(f"{i}:0:".encode() + mpk).digest()).digest()
        assert int.from_bytes(k, "big") == (master + int.from_bytes(seq, "big")) % pyref.N
