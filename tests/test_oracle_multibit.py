"""Pin the oracle's MultiBit-HD restatement: the reference's issue-445 vectors (src/multibit.rs:404-417,
462-479, 367-371) when the BIP-39 list is readable from the reference tree (this container only), and an
independent hashlib/hmac transcription on a synthetic word list everywhere."""
import hashlib
import hmac
import os
import random

import pytest

from tests import pyref

REF_WORDLIST = "/root/reference/src/data/bip39_english.txt"


def synthetic_wordlist(seed=39):
    rnd = random.Random(seed)
    words = set()
    while len(words) < 2048:
        words.add("".join(rnd.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rnd.randrange(3, 9))))
    return sorted(words)


def py_multibit(words, mnemonic, count, passphrase=b""):
    toks = mnemonic.split()
    if len(toks) not in (12, 15, 18, 21, 24) or any(t not in words for t in toks):
        return None
    seed1 = hashlib.pbkdf2_hmac("sha512", mnemonic.encode(), b"mnemonic" + passphrase, 2048)
    bits = "".join(f"{b:08b}" for b in seed1) + "".join(f"{b:08b}" for b in hashlib.sha256(seed1).digest())[:16]
    buggy = " ".join(words[int(bits[11 * i:11 * i + 11], 2)] for i in range(48))
    seed2 = hashlib.pbkdf2_hmac("sha512", buggy.encode(), b"mnemonic" + passphrase, 2048)
    I = hmac.new(b"Bitcoin seed", seed2, hashlib.sha512).digest()
    N = pyref.N

    def ckd(k, c, idx, hard):
        ki = int.from_bytes(k, "big")
        if not 1 <= ki < N:
            return None
        data = (b"\0" + k if hard else pyref.pubkeys(ki)[0]) + ((idx | 0x80000000) if hard else idx).to_bytes(4, "big")
        I = hmac.new(c, data, hashlib.sha512).digest()
        il = int.from_bytes(I[:32], "big")
        if il >= N or (il + ki) % N == 0:
            return None
        return ((il + ki) % N).to_bytes(32, "big"), I[32:]

    out = []
    a = ckd(I[:32], I[32:], 0, True)
    b = ckd(*a, 0, False) if a else None
    if b:
        for i in range(count):
            r = ckd(*b, i, False)
            if r:
                out.append(r[0])
    return seed1, seed2, buggy, out


def test_synthetic_wordlist_vs_python(vko):
    words = synthetic_wordlist()
    wl = vko.Wordlist(words)
    rnd = random.Random(40)
    cases = [" ".join(rnd.choice(words) for _ in range(n)) for n in (12, 15, 18, 21, 24, 12, 24)]
    cases += ["  " + cases[0].replace(" ", "\t", 3) + " \n", " ".join(words[:11]), " ".join(words[:12]) + " notaword", ""]
    for m in cases:
        exp = py_multibit(words, m, 5)
        got = vko.multibit_keys(wl, m, count=5)
        if exp is None:
            assert got is None, m
            continue
        seed1, seed2, buggy, keys = exp
        assert (got[0], got[1]) == (seed1, seed2), m
        assert vko.multibit_buggy_mnemonic(wl, seed1) == buggy
        assert [k for _, k in got[2]] == keys
    # BIP-39 passphrase (multibit-hd analyzer, src/analyze/multibit.rs:47-52): salt = "mnemonic" || passphrase, both stretches
    for pw in (b"TREZOR", b"x" * 96, "z\u00fc".encode(), b"y" * 257, bytes(range(1, 256)) * 5):     # any length (round 5)
        seed1, seed2, _, keys = py_multibit(words, cases[0], 3, pw)
        got = vko.multibit_keys(wl, cases[0], count=3, passphrase=pw)
        assert (got[0], got[1], [k for _, k in got[2]]) == (seed1, seed2, keys), pw
    toks = cases[0].split()
    assert vko.multibit_keys(wl, cases[0], count=2)[2][1][0] == f"{toks[0]} {toks[1]}...{toks[-2]} {toks[-1]}[m/0'/0/1]"


@pytest.mark.skipif(not os.path.exists(REF_WORDLIST), reason="BIP-39 list only exists inside the reference tree")
def test_reference_issue_445_vectors(vko):
    words = open(REF_WORDLIST).read().split()
    assert len(words) == 2048 and words[0] == "abandon" and words[2047] == "zoo"      # src/multibit.rs:367-371
    wl = vko.Wordlist(words)
    m = "skin join dog sponsor camera puppy ritual diagram arrow poverty boy elbow"
    seed1, seed2, keys = vko.multibit_keys(wl, m)
    buggy = vko.multibit_buggy_mnemonic(wl, seed1).split()
    assert len(buggy) == 48 and buggy[:3] == ["trim", "snack", "gorilla"] and buggy[47] == "coach"   # :462-479
    assert len(keys) == 20 and keys[0][0] == "skin join...boy elbow[m/0'/0/0]"
    assert vko.derive(keys[0][1]).p2pkh_compressed == b"1LQ8XnNKqC7Vu7atH5k4X8qVCc9ug2q7WE"          # :404-417
    assert keys[0][1].hex() == "9e143561ac6d038bb7d27321dfd56286a21ea4c01c2b1d2fdfb41380beaa5409"
    assert vko.multibit_keys(wl, "abandon abandon abandon") is None                                    # :426-433
    assert vko.multibit_keys(wl, "abandon " * 11 + "notaword") is None                                 # :435-443
    assert vko.multibit_keys(wl, "abandon " * 11 + "about") is not None                                # :419-424


def issue445_fixture():
    import json
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multibit_issue445.json")))
    words = []
    for i in range(2048):  # the fixture's filler rule
        s, v = "", i
        for _ in range(3):
            s = chr(ord("a") + v % 26) + s
            v //= 26
        words.append("q" + s)
    for i, w in fx["words_at"]:
        words[i] = w
    return fx, words


def test_issue_445_fixture_without_the_reference_tree(vko):
    """tests/golden/multibit_issue445.json (made from the real list by make_issue445.py) reproduces the reference vector
    through the oracle with only the ~58 touched positions of the list filled in: this is what the GPU test replays."""
    fx, words = issue445_fixture()
    assert len(set(words)) == 2048
    seed1, seed2, keys = vko.multibit_keys(vko.Wordlist(words), fx["mnemonic"])
    assert (seed1.hex(), seed2.hex()) == (fx["seed1"], fx["seed2"])
    assert [[l, k.hex()] for l, k in keys] == fx["keys"] and len(keys) == 20
    assert keys[0][1].hex() == "9e143561ac6d038bb7d27321dfd56286a21ea4c01c2b1d2fdfb41380beaa5409"
    assert vko.derive(keys[0][1]).p2pkh_compressed == b"1LQ8XnNKqC7Vu7atH5k4X8qVCc9ug2q7WE" == fx["key0_p2pkh_compressed"].encode()
