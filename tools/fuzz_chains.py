#!/usr/bin/env python3
"""Randomised parity soak of the key-derivation chains on the GPU box: batches of random size (so that the ring sees 1 .. ~80
groups, partial last groups, worker counts on both sides of the group count), mnemonics with random word counts / white space /
invalid words and hundreds of redundant blanks (lines on both sides of the 448-byte staging slice), Electrum seeds of every length
0 .. 1200 (fast path, long path) and non-hex lines; the device's
keys against the CPU oracle for a random sample of every batch.  usage: tools/fuzz_chains.py [batches] [seed]"""
import json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vuke_amd
from vuke_amd import _ffi as ffi
from oracle import vko
from tests.test_oracle_multibit import synthetic_wordlist

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
vko.build()
words = synthetic_wordlist()
wl = vko.Wordlist(words)
eng = vuke_amd.Engine(0, window_bits=16)
eng.bip39_wordlist_set(words)
WS = [" ", "  ", "\t", " \t ", "\n", "\x0b", "\x0c", "\r "]
bad = 0
t0 = time.time()

def mnemonic():
    r = rnd.random()
    n = rnd.choice([12, 15, 18, 21, 24]) if r < 0.8 else rnd.choice([0, 1, 11, 13, 25])
    ws = [rnd.choice(words) for _ in range(n)]
    if r > 0.9 and ws:
        ws[rnd.randrange(len(ws))] = rnd.choice(["notaword", "zzzzzzzzz", "a", words[3][:-1] or "x"])
    sep = (lambda: " ") if rnd.random() < 0.7 else (lambda: rnd.choice(WS))
    if rnd.random() < 0.15:   # a line padded past the staging slice: its HMAC key is hashed from the input bytes
        pad = rnd.choice([20, 36, 40, 60, 120])
        sep = lambda: " " * rnd.randrange(1, pad)
    s = (rnd.choice(["", " ", "\t"]) if rnd.random() < 0.2 else "") + "".join(w + (sep() if i + 1 < len(ws) else "") for i, w in enumerate(ws)) + (rnd.choice(["", " ", "\n"]))
    return s.encode()

def eseed():
    r = rnd.random()
    n = rnd.choice([0, 1, 22, 23, 24, 32, 55, 56, 63, 64, 65, 66, 128, 200, 447, 448, 449, 460, 535, 536, 1000, 1200]) if r < 0.5 else rnd.randrange(0, 100)
    s = "".join(rnd.choice("0123456789abcdefABCDEF") for _ in range(n))
    if rnd.random() < 0.1 and n:
        i = rnd.randrange(n)
        s = s[:i] + rnd.choice("gz -_") + s[i + 1:]
    return s.encode()

for b in range(batches):
    n = rnd.choice([1, 2, 63, 64, 65, 128, 129, rnd.randrange(1, 400), rnd.randrange(1, 5000)])
    count = rnd.choice([1, 2, 20])
    if b % 2 == 0:
        items = [mnemonic() for _ in range(n)]
        rows = eng.transform_strings(ffi.spec(ffi.T_MULTIBIT, 0, 0, count), items)
        check = lambda m: (lambda e: [k for _, k in e[2]] if e else [])(vko.multibit_keys(wl, m, count=count))
        name = "multibit"
    else:
        change = rnd.randrange(2)
        items = [eseed() for _ in range(min(n, 600))]
        rows = eng.transform_strings(ffi.spec(ffi.T_ELECTRUM, change, 0, count), items)
        check = lambda s: [k for _, k in vko.electrum_keys(s, for_change=bool(change), count=count)[1]]
        name = "electrum"
    sample = rnd.sample(range(len(items)), min(len(items), 24 if name == "multibit" else 8))
    ok = True
    for i in sample:
        got = [k for k in rows[i] if k is not None]
        want = check(items[i])
        if got != want:
            ok = False
            print(json.dumps({"batch": b, "kind": name, "index": i, "input": items[i].decode("latin1"), "got": len(got), "want": len(want)}), flush=True)
    bad += not ok
    print(json.dumps({"batch": b, "kind": name, "n": len(items), "count": count, "ok": ok, "elapsed": round(time.time() - t0, 1)}), flush=True)
print(json.dumps({"batches": batches, "failures": bad}))
sys.exit(1 if bad else 0)
