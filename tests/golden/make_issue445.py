#!/usr/bin/env python3
"""Generate tests/golden/multibit_issue445.json: the reference's MultiBit-HD issue-445 vector
(src/multibit.rs:404-417, :462-479) as a fixture that travels to the GPU box without the BIP-39 file.

The chain touches at most 60 list positions: the 12 words of the mnemonic (membership test) and the 48 words rebuilt
from seed1 (index -> word).  The fixture holds exactly those (index, word) pairs; every other position of the
2048-entry list is filled by the rule filler(i) = "q" + base-26 letters of i (no English word starts with q + such a
tail of consonants in the touched set; the script asserts no collision).  Expected values -- seed1, seed2, the 48
words, the 20 child keys, the address of key 0 -- come from the oracle run on the REAL list here, and the script asserts
the sparse list reproduces them.  Runs only where /root/reference exists (the build container).

    python tests/golden/make_issue445.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import vko  # noqa: E402

REF_WORDLIST = "/root/reference/src/data/bip39_english.txt"
MNEMONIC = "skin join dog sponsor camera puppy ritual diagram arrow poverty boy elbow"


def filler(i):
    s = ""
    for _ in range(3):
        s = chr(ord("a") + i % 26) + s
        i //= 26
    return "q" + s


def sparse_list(pairs):
    words = [filler(i) for i in range(2048)]
    for i, w in pairs:
        words[int(i)] = w
    return words


def main():
    real = open(REF_WORDLIST).read().split()
    assert len(real) == 2048
    wl = vko.Wordlist(real)
    seed1, seed2, keys = vko.multibit_keys(wl, MNEMONIC)
    buggy = vko.multibit_buggy_mnemonic(wl, seed1).split()
    assert len(buggy) == 48 and buggy[:3] == ["trim", "snack", "gorilla"] and buggy[47] == "coach"
    touched = sorted({real.index(w) for w in MNEMONIC.split()} | {real.index(w) for w in buggy})
    pairs = [[i, real[i]] for i in touched]
    words = sparse_list(pairs)
    assert len(set(words)) == 2048 and all(len(w) <= 8 for w in words)
    s1, s2, k2 = vko.multibit_keys(vko.Wordlist(words), MNEMONIC)
    assert (s1, s2, k2) == (seed1, seed2, keys)
    d0 = vko.derive(keys[0][1])
    assert d0.p2pkh_compressed == b"1LQ8XnNKqC7Vu7atH5k4X8qVCc9ug2q7WE"
    out = {
        "what": "MultiBit-HD issue-445 vector of the reference (src/multibit.rs:404-417), sparse word list",
        "generator": "tests/golden/make_issue445.py",
        "mnemonic": MNEMONIC,
        "words_at": pairs,
        "filler_rule": "'q' + three base-26 letters of the index, most significant first",
        "seed1": seed1.hex(), "seed2": seed2.hex(), "buggy_mnemonic": " ".join(buggy),
        "keys": [[label, k.hex()] for label, k in keys],
        "key0_p2pkh_compressed": d0.p2pkh_compressed.decode(),
        "key0_h160_c": bytes(d0.h160_c).hex(),
        "key0_h160_u": bytes(d0.h160_u).hex(),
    }
    json.dump(out, open(os.path.join(HERE, "multibit_issue445.json"), "w"), indent=0)
    print("wrote multibit_issue445.json:", len(pairs), "real words")


if __name__ == "__main__":
    main()
