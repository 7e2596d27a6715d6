"""Synthetic inputs of the BASELINE.json configs, defined once for bench.py, the GPU tests and the
fixture generator (tests/golden/make_bench_fixtures.py).  Pure data generation: no device, no oracle.

  config 2  Milksad seeds 0..2^32-1, compressed P2PKH; 1024 (+2) planted seeds
  config 3  sha256 transform (3 keys / input), both hash160s; 1 000 000 targets = 1 000 planted + 999 000 decoys
  config 4  all four LCG variants, seeds 0..2^31-1, compressed; 256 planted per variant
  config 5  100 000 synthetic 12-word mnemonics (MultiBit-HD) / 100 000 32-hex-digit seeds (Electrum v1), 20 keys each;
            planted = key 7 of every 1000th input
"""
import hashlib
import json
import os
import struct

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")
MASK64 = 2**64 - 1


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    return json.load(open(path)) if os.path.exists(path) else None


# ---------------------------------------------------------------- digest (DESIGN.md "sweep digest")
def fold_digests(parts):
    d = [0, 0, 0, 0]
    for p in parts:
        for i in range(3):
            d[i] ^= p[i]
        d[3] = (d[3] + p[3]) & MASK64
    return tuple(d)


def digest_of_h160s(h160s):
    """The device's / oracle's digest over a list of 20-byte hash160s, in python (fixtures and tests)."""
    d = [0, 0, 0, 0]
    for h in h160s:
        w = struct.unpack("<5I", h)
        a, b = w[0] | (w[1] << 32), w[2] | (w[3] << 32)
        d[0] ^= a
        d[1] ^= b
        d[2] ^= w[4]
        d[3] = (d[3] + a + b + w[4]) & MASK64
    return tuple(d)


def hexd(d):
    return [f"{x:016x}" for x in d]


def unhexd(d):
    return tuple(int(x, 16) for x in d)


# ---------------------------------------------------------------- config 2
C2_BLOCK_LOG2 = 28


def c2_planted_seeds():
    return sorted({(0x9E3779B1 * j + 12345) & 0xFFFFFFFF for j in range(1024)} | {0, 2**32 - 1})


# ---------------------------------------------------------------- config 3
C3_INPUTS = 100_000_000
C3_STEP_INPUTS = 1 << 25
C3_WARMUP_INPUTS = 1 << 18


def c3_planted():
    """(seed, sub_index, kind): kind 0 p2pkh(compressed key), 1 p2pkh(uncompressed key), 2 p2wpkh."""
    return [(7919 * j, j % 3, j % 3) for j in range(1000)]


def c3_decoys(n=999_000):
    """(hash160, target kind): first 20 bytes of SHA256("decoy" || LE64(j)), kind = j mod 2 (0 p2pkh, 1 p2wpkh)."""
    return [(hashlib.sha256(b"decoy" + struct.pack("<Q", j)).digest()[:20], j & 1) for j in range(n)]


# ---------------------------------------------------------------- config 4
C4_SEEDS = 1 << 31
C4_STEP_SEEDS = 1 << 26
C4_WARMUP_SEEDS = 1 << 18


def c4_planted():
    """(seed, variant): 256 per variant; minstd accepts seeds <= 2^31 - 2 only (src/lcg.rs:77-83)."""
    out = []
    for v in range(4):
        for j in range(256):
            s = (0x9E3779B1 * j + 777 + 1000003 * v) % (2**31 - 1)
            out.append((s, v))
    return out


# ---------------------------------------------------------------- config 5
C5_INPUTS = 100_000
C5_KEYS_PER_INPUT = 20
C5_PLANT_EVERY = 1000
C5_PLANT_INDEX = 7
C5_WARMUP_MULTIBIT = 256
C5_WARMUP_ELECTRUM = 64


def c5_wordlist():
    """The 2048-word list of the config-5 runs: the committed sparse issue-445 list (tests/golden/multibit_issue445.json:
    real BIP-39 words at 58 positions, the rule 'q' + base-26 index elsewhere).  The reference does not verify the
    checksum (src/multibit.rs:155-157), so any 2048 distinct words exercise the same chain."""
    fx = load_golden("multibit_issue445.json")
    words = []
    for i in range(2048):
        s, v = "", i
        for _ in range(3):
            s = chr(ord("a") + v % 26) + s
            v //= 26
        words.append("q" + s)
    for i, w in fx["words_at"]:
        words[i] = w
    return words


def c5_mnemonics(n=C5_INPUTS, first=0):
    """Mnemonic i = 12 words indexed by outputs 12 i .. 12 i + 11 of MT19937(seed 5489) mod 2048."""
    import numpy as np
    words = c5_wordlist()
    rs = np.random.RandomState(5489)
    idx = rs.randint(0, 2**32, size=12 * (first + n), dtype=np.uint64)[12 * first:] % 2048
    idx = idx.reshape(n, 12)
    return [" ".join(words[j] for j in row).encode() for row in idx]


def c5_electrum_seeds(n=C5_INPUTS, first=0):
    """Seed i = lowercase hex of bytes 16 i .. 16 i + 15 of numpy RandomState(5489).bytes (MT19937 output, LE words)."""
    import numpy as np
    raw = np.random.RandomState(5489).bytes(16 * (first + n))[16 * first:]
    return [raw[16 * i:16 * i + 16].hex().encode() for i in range(n)]


def c5_planted_inputs(n=C5_INPUTS):
    return list(range(0, n, C5_PLANT_EVERY))
