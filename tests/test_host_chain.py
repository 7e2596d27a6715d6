"""The key-derivation chain kernels (vk_chain.cuh, vk_multibit.cuh, vk_electrum.cuh) on a machine without a GPU: the cold
kernels and the chunk bodies compiled for the host one lane at a time (tests/host_emu/host_chain_emu.cpp shims HIP's kernel
syntax and the gfx950 builtins), held to hashlib -- PBKDF2-HMAC-SHA512, SHA-256, SHA-512 -- and to a python restatement of
the BIP-39 repacking.  Checks the logic of the staged strings, the padding patched in registers, the salt blocks, the word
extraction and the ring's bookkeeping; memory ordering between waves is what the GPU parity tests check."""
import hashlib
import os
import random
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"   # the sources use clang's ext_vector_type

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs ROCm's clang++ for the host build")


def words2048():
    out = []
    for i in range(2048):
        s, v = "", i
        for _ in range(3):
            s = chr(ord("a") + v % 26) + s
            v //= 26
        out.append("w" + s + ("xyzq"[: i % 5]))   # 4 .. 8 letters
    return out


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    d = tmp_path_factory.mktemp("chain_emu")
    exe = d / "chain_emu"
    subprocess.run([CLANG, "-std=c++17", "-O2", "-Wno-unknown-pragmas", "-Wno-unknown-attributes", "-I", os.path.join(ROOT, "vuke_amd", "csrc"),
                    "-I", os.path.join(ROOT, "tests", "host_emu"), "-o", str(exe), os.path.join(ROOT, "tests", "host_emu", "host_chain_emu.cpp")],
                   check=True, capture_output=True, text=True)
    wl = d / "words.txt"
    wl.write_text("\n".join(words2048()) + "\n")

    def run(lines):
        r = subprocess.run([str(exe), str(wl)], input="".join(l + "\n" for l in lines), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-500:])
        return r.stdout.splitlines()
    return run


def hx(b):
    return b.hex() if b else "-"


def test_staged_hashes_against_hashlib(emu):
    rnd = random.Random(7)
    lens = [0, 1, 3, 4, 7, 8, 9, 55, 56, 63, 64, 65, 111, 112, 119, 120, 127, 128, 129, 239, 240, 255, 256, 300, 431, 447, 448, 495]
    msgs = [bytes(rnd.randrange(256) for _ in range(n)) for n in lens]
    got = emu([f"sha512stage {hx(m)}" for m in msgs])
    assert [g.strip() for g in got] == [hashlib.sha512(m).hexdigest() for m in msgs]
    msgs = [m for m in msgs if len(m) <= 503]
    got = emu([f"sha256stage {hx(m)}" for m in msgs])
    assert [g.strip() for g in got] == [hashlib.sha256(m).hexdigest() for m in msgs]


def repack48(seed1, words):
    bits = int.from_bytes(seed1 + hashlib.sha256(seed1).digest()[:2], "big")
    return " ".join(words[(bits >> (11 * (47 - k))) & 0x7FF] for k in range(48))


def test_multibit_prepare_stretch_and_repacking(emu):
    words = words2048()
    rnd = random.Random(11)
    cases = []
    for n in (12, 15, 24):
        cases.append((" ".join(rnd.choice(words) for _ in range(n)), b""))
    cases.append(("  " + "\t\t".join(rnd.choice(words) for _ in range(12)) + " \n", b"TREZOR"))            # password = the string as given
    cases.append(((" " * 30).join(rnd.choice(words) for _ in range(12)), b"p" * 96))                        # key > 128 bytes, one salt block
    # round 5: a BIP-39 passphrase of any length (96 bytes at most before): around the block boundaries of the salt message
    for pl in (97, 99, 100, 227, 228, 1000):
        cases.append((" ".join(rnd.choice(words) for _ in range(12)), bytes(rnd.randrange(1, 256) for _ in range(pl))))
    # beyond the 448 bytes of the staging slice: the HMAC key is hashed from the input bytes (sha512_bytes); lengths around the slice
    # limit and the SHA-512 block / padding boundaries (a 12-word mnemonic padded with blanks between and behind its words)
    base = (" " * 33).join(rnd.choice(words) for _ in range(12))
    for total in (448, 449, 495, 496, 511, 512, 623, 624, 1000, 3000):
        cases.append((base + " " * (total - len(base)), b"pw"))
    bad = [" ".join(words[:11]), " ".join(words[:12]) + " zzzzzzzzz", " ".join(words[:11]) + " " + words[11][:-1], "",
           " ".join(words[:11]) + " " + words[11] + "\0", " ".join(words[:13])]
    out = emu([f"multibit {hx(m.encode())} {hx(p)}" for m, p in cases] + [f"multibit {hx(m.encode())} -" for m in bad])
    i = 0
    for m, p in cases:
        assert out[i] == "ok 1", (m, out[i])
        seed1 = hashlib.pbkdf2_hmac("sha512", m.encode(), b"mnemonic" + p, 2048)
        assert out[i + 1] == "seed1 " + seed1.hex()
        w48 = repack48(seed1, words)
        assert out[i + 2] == "words " + w48
        assert out[i + 3] == "seed2 " + hashlib.pbkdf2_hmac("sha512", w48.encode(), b"mnemonic" + p, 2048).hex()
        i += 4
    assert out[i:] == ["ok 0"] * len(bad)


def test_bitimage_prepare_and_stretch(emu):
    words = words2048()
    rnd = random.Random(13)
    lines, want = [], []
    for p in (b"", b"x", b"q" * 100, b"z" * 229, b"s" * 256, b"t" * 257, b"u" * 355, b"v" * 356, b"w" * 5000):     # one, two, three ... forty salt blocks
        ent = bytes(rnd.randrange(256) for _ in range(32))
        bits = int.from_bytes(ent + hashlib.sha256(ent).digest()[:1], "big")
        mn = " ".join(words[(bits >> (11 * (23 - k))) & 0x7FF] for k in range(24))
        lines.append(f"bitimage {ent.hex()} {hx(p)}")
        want += ["words " + mn, "seed " + hashlib.pbkdf2_hmac("sha512", mn.encode(), b"mnemonic" + p, 2048).hex()]
    assert emu(lines) == want


def electrum_stretch(seed, rounds):
    x = hashlib.sha256(seed + seed).digest()
    for _ in range(rounds - 1):
        x = hashlib.sha256(x + seed).digest()
    return x


def test_electrum_prepare_stretch_and_long_seeds(emu):
    # any length (the reference hashes the whole slice, src/electrum.rs:57-78,150-168): beyond 64 bytes the long path reads the
    # input bytes themselves; lengths around every block boundary of x || seed || padding, and past the 448 bytes of rounds 2-3
    seeds = [b"", b"0", b"7f", b"abc", b"f" * 22, b"f" * 23, b"f" * 24, b"0123456789abcdefABCDEF0123456789", b"a" * 55, b"b" * 63, b"c" * 64,
             b"d" * 65, b"e" * 87, b"1" * 88, b"2" * 300, b"3" * 448, b"4" * 449, b"5" * 535, b"6" * 536, b"7" * 1000, b"89abcdef" * 400]
    out = emu([f"electrum {hx(s)} 150" for s in seeds] + ["electrum " + hx(b"xyz") + " 3", "electrum " + hx(b"4" * 500 + b"g") + " 3", f"electrum {hx(b'5' * 32)} 1"])
    for i, s in enumerate(seeds):
        assert out[2 * i] == ("ok 1" if len(s) <= 64 else "ok 2"), s
        assert out[2 * i + 1] == "x " + electrum_stretch(s, 150).hex(), s
    assert out[2 * len(seeds):] == ["ok 0", "ok 0", "ok 1", "x " + electrum_stretch(b"5" * 32, 1).hex()]
    # round 5: the long path takes every word that lies inside the seed from the two aligned dwords around it and runs in slices
    # of a bounded number of rounds (7 here): seeds that start 1..3 bytes into the buffer, lengths of every residue mod 4
    rnd = random.Random(5)
    cases = [(bytes(rnd.choice(b"0123456789abcdefABCDEF") for _ in range(n)), k) for k in (1, 2, 3, 5) for n in (65, 66, 67, 68, 119, 120, 121, 250, 997)]
    out = emu([f"electrum {hx(s)} 23@{k}" for s, k in cases])
    for i, (s, k) in enumerate(cases):
        assert out[2 * i: 2 * i + 2] == ["ok 2", "x " + electrum_stretch(s, 23).hex()], (len(s), k)


def test_ring_bookkeeping_with_one_worker(emu):
    """One worker, G groups, C chunks: tickets come out in FIFO order (every group advances one chunk per lap), each (group,
    chunk) exactly once, and the counters end at head = total + 1 (the pop that found nothing), tail = total."""
    for groups, chunks in ((1, 1), (1, 5), (3, 4), (7, 1), (64, 3), (100, 2)):
        (line,) = emu([f"ring {groups} {chunks}"])
        seq = [tuple(map(int, t.split(":"))) for t in line.split()]
        assert seq == [(g, c) for c in range(chunks) for g in range(groups)]


def test_host_and_device_share_the_kernel_sources():
    """The harness includes the shipped headers; nothing under vuke_amd/csrc knows about it."""
    for f in os.listdir(os.path.join(ROOT, "vuke_amd", "csrc")):
        if f.endswith((".cuh", ".hip", ".h")):
            assert "host_chain_emu" not in open(os.path.join(ROOT, "vuke_amd", "csrc", f)).read()
    assert shutil.which("g++")
