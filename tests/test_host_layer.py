"""CPU tests of the C++ host layer (vuke_amd/host): transform grammar, labels, text codecs, record
assembly, output formats, Matcher -- checked against the oracle and the reference's own tests."""
import ctypes as C
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    import __graft_entry__ as ge
    ge.build()
    L = C.CDLL(os.environ.get("VUKE_HOST_LIB") or os.path.join(ROOT, "vuke_amd", "lib", "libvuke_host.so"))   # VUKE_HOST_LIB: the sanitizer build (tools/run_sanitizers.sh)
    L.vkh_matcher_load.restype = C.c_long
    return L


class Spec(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("variant", C.c_uint32), ("endian", C.c_uint32), ("chain_depth", C.c_uint32)]


ALL = 0xFFFFFFFF


def parse(host, s):
    sp, name, err = Spec(), C.create_string_buffer(64), C.create_string_buffer(512)
    rc = host.vkh_parse_transform(s.encode(), C.byref(sp), name, 64, err, 512)
    if rc != 0:
        return None, err.value.decode()
    return (sp.kind, sp.variant, sp.endian, sp.chain_depth), name.value.decode()


def test_transform_grammar_reference_cases(host):
    # src/lcg.rs parse_tests, src/xorshift.rs:334-357, src/sha256_chain.rs:187-217, src/transform/mod.rs:170-215
    assert parse(host, "sha256") == ((0, ALL, 0, 10), "sha256")
    assert parse(host, "SHA256") == ((0, ALL, 0, 10), "sha256")
    assert parse(host, "milksad") == ((1, ALL, 0, 10), "milksad")
    assert parse(host, "mt64") == ((2, ALL, 0, 10), "mt64")
    assert parse(host, "lcg") == ((3, ALL, 0, 10), "lcg")
    assert parse(host, "lcg:le") == ((3, ALL, 1, 10), "lcg")
    assert parse(host, "lcg:glibc") == ((3, 0, 0, 10), "lcg")
    assert parse(host, "lcg:GLIBC:LE") == ((3, 0, 1, 10), "lcg")
    assert parse(host, "lcg:borland:be") == ((3, 3, 0, 10), "lcg")
    assert parse(host, "lcg:") == ((3, ALL, 0, 10), "lcg")          # trailing ':' trimmed
    assert parse(host, "lcg:big") == ((3, ALL, 0, 10), "lcg")
    assert parse(host, "xorshift") == ((4, ALL, 0, 10), "xorshift")
    for alias, v in (("64", 0), ("xorshift64", 0), ("128", 1), ("128plus", 2), ("128+", 2), ("plus", 2), ("xoroshiro", 3),
                     ("starstar", 3), ("xoroshiro128**", 3), ("xoroshiro128starstar", 3)):
        assert parse(host, "xorshift:" + alias) == ((4, v, 0, 10), "xorshift")
    assert parse(host, "sha256_chain") == ((5, ALL, 0, 10), "sha256_chain")
    assert parse(host, "sha256_chain:iterated") == ((5, 0, 0, 10), "sha256_chain:iterated")
    assert parse(host, "sha256_chain:indexed") == ((5, 1, 0, 10), "sha256_chain:indexed:be")
    assert parse(host, "sha256_chain:indexed:be") == ((5, 1, 0, 10), "sha256_chain:indexed:be")
    assert parse(host, "sha256_chain:indexed:le") == ((5, 2, 0, 10), "sha256_chain:indexed:le")
    assert parse(host, "sha256_chain:counter") == ((5, 3, 0, 10), "sha256_chain:counter")
    # errors carry the reference's wording
    assert parse(host, "lcg:nope") == (None, "Invalid LCG variant or endian: 'nope'. Valid variants: glibc, minstd, msvc, borland. Valid endian: be, le")
    assert parse(host, "lcg:glibc:xx") == (None, "Invalid endian: 'xx'. Valid: be, le")
    assert parse(host, "lcg:a:b:c")[1].startswith("Invalid LCG format")
    assert parse(host, "xorshift:7")[1].startswith("Invalid xorshift variant: '7'")
    assert parse(host, "sha256_chain:foo")[1].startswith("Invalid SHA256 chain variant: 'foo'")
    assert parse(host, "sha256_chain:iterated:le")[1].startswith("Invalid SHA256 chain format")
    assert parse(host, "nothing")[1].startswith("Unknown transform: nothing. Valid: direct, sha256,")
    assert [parse(host, n)[0][0] for n in ("double_sha256", "md5", "direct", "armory")] == [8, 9, 10, 11]
    assert parse(host, "bitimage") == ((12, ALL, 0, 10), "bitimage")              # blob inputs only: vk_bitimage_keys
    assert parse(host, "electrum") == ((6, 0, 0, 20), "electrum")                 # src/transform/mod.rs:180-181
    assert parse(host, "electrum:change") == ((6, 1, 0, 20), "electrum:change")
    assert parse(host, "multibit") == ((7, 0, 0, 20), "multibit")                 # src/transform/mod.rs:178


def test_labels_match_oracle(host, vko):
    seeds = [0, 5, 2**31 - 1, 2**32, 2**64 - 1]
    for kind, var, en, depth in [(0, ALL, 0, 10), (1, ALL, 0, 10), (2, ALL, 0, 10), (3, ALL, 0, 10), (3, ALL, 1, 10), (3, 2, 1, 10),
                                 (4, ALL, 0, 10), (4, 3, 0, 10), (5, ALL, 0, 10), (5, ALL, 0, 3), (5, 2, 0, 10)]:
        sp, osp = Spec(kind, var, en, depth), vko.spec(kind, var, en, depth)
        for s in seeds:
            for sub in range(vko.slots_per_input(osp)):
                out = C.create_string_buffer(128)
                host.vkh_label(C.byref(sp), C.c_uint64(s), sub, out, 128)
                assert out.value.decode() == vko.transform_label(osp, s, sub), (kind, var, en, depth, s, sub)


def test_codecs_match_oracle(host, vko):
    rnd = random.Random(21)
    for _ in range(300):
        n = rnd.choice([1, 21, 33, 34, 21, 21])
        p = bytes([rnd.choice([0, 0, rnd.getrandbits(8)])]) + bytes(rnd.getrandbits(8) for _ in range(n - 1))
        out = C.create_string_buffer(128)
        host.vkh_base58check_encode(p, len(p), out, 128)
        assert out.value.decode() == vko.base58check_encode(p)
        dec = C.create_string_buffer(64)
        m = host.vkh_base58check_decode(out.value, dec, 64)
        assert dec.raw[:m] == p
        h = bytes(rnd.getrandbits(8) for _ in range(20))
        host.vkh_segwit_encode(b"bc", 0, h, 20, out, 128)
        assert out.value.decode() == vko.segwit_encode("bc", 0, h)
        ver = C.c_int()
        m = host.vkh_segwit_decode(b"bc", out.value, C.byref(ver), dec, 64)
        assert (ver.value, dec.raw[:m]) == (0, h)
        msg = bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(0, 150)))
        d = C.create_string_buffer(32)
        host.vkh_sha256(msg, len(msg), d)
        assert d.raw == vko.sha256(msg)
    # leading zero bytes, tiny values and every payload length the 58^5-radix encoder splits differently
    out = C.create_string_buffer(128)
    for n in range(1, 41):
        for z in (0, 1, min(n, 3), n):
            p = bytes(z) + bytes(rnd.getrandbits(8) | 1 for _ in range(n - z))
            host.vkh_base58check_encode(p, len(p), out, 128)
            assert out.value.decode() == vko.base58check_encode(p), (n, z)
    dec = C.create_string_buffer(64)
    assert host.vkh_base58check_decode(b"1NonExistentAddress", dec, 64) == -1
    assert host.vkh_base58check_decode(b"1JwSSubhmg6iPtRjtyqhUYYH7bZg3Lfy1U", dec, 64) == -1
    ver = C.c_int()
    assert host.vkh_segwit_decode(b"bc", b"bc1qw508d6qejxtdg4y5r3zarvary0c5xw7kv8f3t5", C.byref(ver), dec, 64) == -1


def _render(host, tmp_path, d, mode, kind=0, source="src", transform="tf"):
    path = str(tmp_path / f"o{mode}{kind}.txt")
    assert host.vkh_render(bytes(d.raw), bytes(d.h160_c), bytes(d.h160_u), mode, kind, source.encode(), transform.encode(), path.encode()) == 0
    return open(path).read()


def test_record_and_console_formats(host, vko, tmp_path):
    rnd = random.Random(22)
    keys = [bytes(32), b"\xff" * 32, bytes.fromhex("c4bbcb1fbec99d65bf59d85c8cb62ee2db963f0fe106f483d9afa73bd4e39a8a")]
    keys += [rnd.getrandbits(rnd.choice([256, 200, 9])).to_bytes(32, "big") for _ in range(12)]
    for k in keys:
        d = vko.derive(k)
        s = lambda b: b.decode()
        rec = _render(host, tmp_path, d, 3).split("|")
        assert rec == [s(d.private_key_decimal), s(d.private_key_binary), str(d.bit_length), str(d.hamming_weight), str(d.leading_zeros),
                       s(d.wif_compressed), s(d.wif_uncompressed), s(d.p2pkh_compressed), s(d.p2pkh_uncompressed), s(d.p2wpkh)]
        # src/output/console.rs:77-81
        assert _render(host, tmp_path, d, 0, source="12345", transform="sha256") == \
            f"12345,sha256,{s(d.private_key_hex)},{s(d.p2pkh_compressed)}\n"
        # src/output/console.rs:66-74
        assert _render(host, tmp_path, d, 1) == (
            f"---\nsource: src\ntransform: tf\nprivate_key: {s(d.private_key_hex)}\nwif_compressed: {s(d.wif_compressed)}\n"
            f"wif_uncompressed: {s(d.wif_uncompressed)}\np2pkh_compressed: {s(d.p2pkh_compressed)}\n"
            f"p2pkh_uncompressed: {s(d.p2pkh_uncompressed)}\np2wpkh: {s(d.p2wpkh)}\n")
        # src/output/console.rs:96-113
        for kind, (addr, nm) in enumerate([(d.p2pkh_compressed, "p2pkh_compressed"), (d.p2pkh_uncompressed, "p2pkh_uncompressed"),
                                           (d.p2wpkh, "p2wpkh")]):
            assert _render(host, tmp_path, d, 2, kind) == (
                f"\n========== HIT ==========\nSource: src\nTransform: tf\nMatched: {s(addr)} ({nm})\n---\n"
                f"Private Key: {s(d.private_key_hex)}\nWIF (compressed): {s(d.wif_compressed)}\nWIF (uncompressed): {s(d.wif_uncompressed)}\n"
                f"---\nP2PKH (compressed): {s(d.p2pkh_compressed)}\nP2PKH (uncompressed): {s(d.p2pkh_uncompressed)}\nP2WPKH: {s(d.p2wpkh)}\n"
                f"=========================\n")
    # reference KAT through the C++ record builder (src/derive.rs:190-197)
    d = vko.derive(keys[2])
    assert "5KJvsngHeMpm884wtkJNzQGaCErckhHJBGFsvd3VyK5qMZXj3hS" in _render(host, tmp_path, d, 3)
    assert "1JwSSubhmg6iPtRjtyqhUYYH7bZg3Lfy1T" in _render(host, tmp_path, d, 3)


def test_matcher_load_check_and_device_form(host, vko, tmp_path):
    d = vko.derive(bytes.fromhex("c4bbcb1fbec99d65bf59d85c8cb62ee2db963f0fe106f483d9afa73bd4e39a8a"))
    c, u, w = d.p2pkh_compressed.decode(), d.p2pkh_uncompressed.decode(), d.p2wpkh.decode()
    f = tmp_path / "t.txt"

    def check(lines):
        f.write_text("\n".join(lines) + "\n")
        return host.vkh_matcher_check(str(f).encode(), bytes(d.raw), bytes(d.h160_c), bytes(d.h160_u))

    assert check([u]) == 1                      # src/matcher.rs:117-138
    assert check(["1NonExistentAddress"]) == -1  # src/matcher.rs:141-149
    assert check([w, u, c]) == 0 and check([w, u]) == 1 and check([w]) == 2 and check([w.upper()]) == -1
    lines = ["  " + c + " \r", "", "   ", "# " + u, c, "junk line", w, w.upper(), "3J98t1WpEZ73CNmQviecrnyiWrnqRhWNLy"]
    f.write_text("\n".join(lines) + "\n")
    h160 = C.create_string_buffer(20 * 16)
    kind = C.create_string_buffer(16)
    n = C.c_size_t()
    cnt = host.vkh_matcher_load(str(f).encode(), h160, kind, 16, C.byref(n))
    assert cnt == 5                              # unique trimmed non-comment lines (count() counts them all)
    got = sorted((kind.raw[i], h160.raw[20 * i:20 * i + 20]) for i in range(n.value))
    assert got == sorted([(0, bytes(d.h160_c)), (1, bytes(d.h160_c))])  # junk, upper-case bech32 and P2SH never match
    assert host.vkh_matcher_load(b"/nonexistent/file", h160, kind, 16, C.byref(n)) == -1


def test_analyze_key_and_cascade_parsers(host):
    # src/analyze/key_parser.rs tests (:232-400)
    def key(s):
        out = C.create_string_buffer(32)
        return out.raw if host.vkh_parse_key(s.encode(), out) == 0 else None

    chbs = "c4bbcb1fbec99d65bf59d85c8cb62ee2db963f0fe106f483d9afa73bd4e39a8a"
    assert key(chbs).hex() == chbs and key("0x" + chbs).hex() == chbs
    assert key("0x15") == bytes(31) + b"\x15" and key("0x202")[30:] == b"\x02\x02" and key("0xd2c55")[29:] == b"\x0d\x2c\x55"
    assert key("1f") == bytes(31) + b"\x1f"
    assert key("5KJvsngHeMpm884wtkJNzQGaCErckhHJBGFsvd3VyK5qMZXj3hS").hex() == chbs
    assert key("L3p8oAcQTtuokSCRHQ7i4MhjWc9zornvpJLfmg62sYpLRJF9woSu").hex() == chbs
    assert key("1") == bytes(31) + b"\x01"
    n_minus_1 = "115792089237316195423570985008687907852837564279074904382605163141518161494336"
    assert int.from_bytes(key(n_minus_1), "big") == int(n_minus_1)
    assert key("115792089237316195423570985008687907852837564279074904382605163141518161494337") is None
    assert key("not a key") is None and key("") is None and key("zzzz") is None

    def casc(s):
        bits, tg, err = C.create_string_buffer(16), (C.c_uint64 * 16)(), C.create_string_buffer(256)
        n = host.vkh_parse_cascade(s.encode(), bits, tg, 16, err, 256)
        return [(bits.raw[i], tg[i]) for i in range(n)] if n >= 0 else err.value.decode()

    assert casc("5:0x15,10:0x202") == [(5, 0x15), (10, 0x202)]
    assert casc("5:21,10:514") == [(5, 21), (10, 514)]
    assert casc("5:0x15,10:514,20:0xd2c55") == [(5, 0x15), (10, 514), (20, 0xd2c55)]
    assert casc("20:0xd2c55,5:0x15,10:0x202") == [(5, 0x15), (10, 0x202), (20, 0xd2c55)]   # sorted by bits
    assert "at least 2 targets" in casc("5:0x15") and "empty input" in casc("  ")
    assert "exceeds 5-bit maximum" in casc("5:0x25,10:0x202") and "bits must be 1-64" in casc("0:1,5:0x15")
    assert "duplicate target" in casc("5:0x15,5:0x15") and "expected 'bits:target'" in casc("5-0x15,10:0x202")


def test_fast_analyzers_run_without_a_device():
    """`vuke analyze KEY --fast` = direct + heuristic (src/analyze/{direct,heuristic}.rs and their tests): key bytes only."""
    import json
    import subprocess
    exe = os.environ.get("VUKE_HIP_EXE") or os.path.join(ROOT, "vuke_amd", "lib", "vuke-hip")
    run = lambda *a: subprocess.run([exe, "analyze", *a], capture_output=True, text=True, timeout=60)
    r = run("0x2a", "--fast")                                   # direct.rs test_small_number
    assert r.returncode == 0, r.stderr
    assert r.stdout == ("Private Key: " + "0" * 62 + "2a\nBit Length:  6\nHamming Weight: 3\n---\nAnalysis:\n"
                        "  ? direct: POSSIBLE (bit_length=6, fits in u64, BE padding detected (31 leading zero bytes))\n"
                        "  ? heuristic: POSSIBLE (low entropy (0.20), unusual hamming weight (3))\n")
    r = run(b"test".hex() + "00" * 28, "--fast", "--json")      # direct.rs test_ascii_string
    d = json.loads(r.stdout)
    assert d["private_key"] == b"test".hex() + "00" * 28 and d["bit_length"] == 255 and d["leading_zeros"] == 0
    assert d["results"][0] == {"analyzer": "direct", "status": "possible", "details": 'LE padding detected (28 trailing zero bytes), ASCII string: "test"'}
    r = run("ab" * 32, "--fast")                                 # one repeated byte is not a "pattern"; 0xab has 5 bits set
    assert "  \u2717 direct: NOT_FOUND (no direct patterns detected)\n" in r.stdout
    assert "  ? heuristic: POSSIBLE (low entropy (0.00), unusual hamming weight (160))\n" in r.stdout
    r = run("0123456789abcdef" * 4, "--fast")                    # heuristic.rs: 8-byte repeating pattern
    assert "repeating byte pattern" in r.stdout and "low entropy (3.00)" in r.stdout
    r = run("c4bbcb1fbec99d65bf59d85c8cb62ee2db963f0fe106f483d9afa73bd4e39a8a", "--fast")
    assert "direct: NOT_FOUND" in r.stdout and "? heuristic: UNKNOWN (entropy=" in r.stdout
    r = run("0x15", "--fast", "--mask", "5")
    assert r.stdout.count("UNKNOWN (masked analysis not supported)") == 2
    r = run("0xff", "--fast", "--mask", "5")
    assert "Warning: key has 8 bits but mask is 5 bits. Key will be treated as already masked." in r.stderr


def test_timestamp_source_dates(host):
    """TimestampSource::from_dates: start 00:00:00 .. end 23:59:59 UTC (src/source/timestamps.rs:21-43), checked against calendar.timegm."""
    import calendar
    host.vkh_timestamp_range.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lo, hi = C.c_uint64(), C.c_uint64()
    for a, b in (("1970-01-01", "1970-01-01"), ("2009-01-03", "2009-01-09"), ("2011-01-01", "2013-12-31"), ("2000-02-29", "2000-03-01"),
                 ("2024-02-29", "2024-02-29"), ("2038-01-19", "2038-01-20"), ("2100-02-28", "2100-03-01")):
        assert host.vkh_timestamp_range(a.encode(), b.encode(), C.byref(lo), C.byref(hi)) == 0
        ya, ma, da = map(int, a.split("-"))
        yb, mb, db = map(int, b.split("-"))
        assert (lo.value, hi.value) == (calendar.timegm((ya, ma, da, 0, 0, 0)), calendar.timegm((yb, mb, db, 23, 59, 59))), (a, b)
    for bad in (("2011-02-29", "2011-03-01"), ("2100-02-29", "2100-03-01"), ("2011-13-01", "2011-12-01"), ("2011-1-1x", "2011-01-02"),
                ("", "2011-01-02"), ("2011-01-02", "2011-01-01")):
        assert host.vkh_timestamp_range(bad[0].encode(), bad[1].encode(), C.byref(lo), C.byref(hi)) == -1, bad


def test_storage_output_takes_ready_made_columns(host, vko, tmp_path):
    """StorageOutput::write_columns (what the device's vk_generate_columns / vk_derive_columns hand over: the value columns as
    Parquet PLAIN bytes) without a GPU: columns built from the oracle's records.  Runs of rows and single records in one sink;
    the files hold the sequence in order, cut into chunk files by record count (exact) and by byte size (within the bounds the
    row sizes give; equality with the record path's cut points is what tests/test_golden.py checks through the CLI), across the
    65 536-row row-group size; malformed columns are refused."""
    import glob
    import struct
    pq = pytest.importorskip("pyarrow.parquet")
    rnd = random.Random(91)
    n_runs, run_rows, singles = 3, 70000, 300
    pool = [vko.derive(rnd.getrandbits(rnd.choice([8, 64, 200, 256])).to_bytes(32, "big") or b"\x01") for _ in range(700)]
    plain = lambda vals: b"".join(struct.pack("<I", len(v)) + v for v in vals)

    def columns(ds, srcs):
        return [plain(srcs), b"".join(bytes(d.raw) for d in ds), plain([d.private_key_hex for d in ds]), plain([d.private_key_decimal for d in ds]),
                plain([d.private_key_binary for d in ds]), b"".join(struct.pack("<I", d.bit_length) for d in ds),
                b"".join(struct.pack("<I", d.hamming_weight) for d in ds), b"".join(struct.pack("<I", d.leading_zeros) for d in ds),
                plain([d.pubkey_compressed for d in ds]), plain([d.pubkey_uncompressed for d in ds]), plain([d.p2pkh_compressed for d in ds]),
                plain([d.p2pkh_uncompressed for d in ds]), plain([d.p2wpkh for d in ds]), plain([d.wif_compressed for d in ds]), plain([d.wif_uncompressed for d in ds])]

    seq, runs = [], []                       # the whole sequence of (source, record) in file order; the runs as column sets
    single_ds = [pool[rnd.randrange(len(pool))] for _ in range(singles)]
    single_src = [f"single{i}".encode() for i in range(singles)]
    nxt = 0
    for b in range(n_runs):
        ds = [pool[rnd.randrange(len(pool))] for _ in range(run_rows)]
        srcs = [f"{b}:{i}".encode() for i in range(run_rows)]
        runs.append(columns(ds, srcs))
        seq += list(zip(srcs, ds))
        upto = singles * (b + 1) // n_runs
        seq += list(zip(single_src[nxt:upto], single_ds[nxt:upto]))
        nxt = upto
    keep = [c for r in runs for c in r]
    data = (C.c_char_p * len(keep))(*keep)
    lens = (C.c_size_t * len(keep))(*[len(c) for c in keep])
    rows = (C.c_uint64 * n_runs)(*[run_rows] * n_runs)
    raw = b"".join(bytes(d.raw) for d in single_ds)
    hc, hu = b"".join(bytes(d.h160_c) for d in single_ds), b"".join(bytes(d.h160_u) for d in single_ds)
    xy = b"".join(bytes(d.pub65)[1:] for d in single_ds)
    host.vkh_storage_write_columns.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_char_p),
                                               C.POINTER(C.c_size_t), C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_char_p]
    host.vkh_storage_write.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                       C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_char_p]
    for label, chunk_records, chunk_bytes in (("count", 100000, 0), ("bytes", 0, 40 << 20)):
        d1 = tmp_path / ("cols_" + label)
        nf = host.vkh_storage_write_columns(str(d1).encode(), b"t", chunk_records, chunk_bytes, n_runs, rows, data, lens, singles, raw, hc, hu, xy,
                                            (C.c_char_p * singles)(*single_src), b"snappy")
        assert nf > 1
        files = sorted(glob.glob(str(d1 / "transform=t" / "date=*" / "chunk_*.parquet")))
        assert len(files) == nf
        got_rows = [pq.read_metadata(f).num_rows for f in files]
        assert sum(got_rows) == len(seq)
        if chunk_records:
            assert got_rows == [100000, 100000, len(seq) - 200000]
            assert pq.read_metadata(files[0]).num_row_groups >= 2
        else:   # every file but the last closes on the first row that takes it to 40 MiB of record data (~890 B per row)
            assert all(44000 < r < 49000 for r in got_rows[:-1])
        t = pq.read_table(files[0]).to_pylist() + pq.read_table(files[-1]).to_pylist()
        want = seq[:got_rows[0]] + seq[len(seq) - got_rows[-1]:]
        assert len(t) == len(want)
        for row, (src, d) in list(zip(t, want))[::97] + list(zip(t, want))[-3:]:
            assert row["source"] == src.decode() and row["private_key_raw"] == bytes(d.raw) and row["private_key_decimal"] == d.private_key_decimal.decode()
            assert row["wif_uncompressed"] == d.wif_uncompressed.decode() and row["address_p2wpkh"] == d.p2wpkh.decode() and row["matched_target"] is None
            assert (row["private_key_bit_length"], row["private_key_hamming_weight"], row["private_key_leading_zeros"]) == (d.bit_length, d.hamming_weight, d.leading_zeros)
            assert row["transform"] == "t" and row["chain"] == "bitcoin" and row["pubkey_uncompressed"] == d.pubkey_uncompressed.decode()
    # a string column that ends inside a value, and a fixed-width column of the wrong size
    bad = list(keep[:15])
    bad[3] = bad[3][:-2]
    assert host.vkh_storage_write_columns(str(tmp_path / "bad1").encode(), b"t", 0, 0, 1, rows, (C.c_char_p * 15)(*bad), (C.c_size_t * 15)(*[len(c) for c in bad]),
                                          0, raw, hc, hu, xy, (C.c_char_p * singles)(*single_src), b"none") == -1
    bad = list(keep[:15])
    bad[1] = bad[1] + bytes(32)
    assert host.vkh_storage_write_columns(str(tmp_path / "bad2").encode(), b"t", 0, 0, 1, rows, (C.c_char_p * 15)(*bad), (C.c_size_t * 15)(*[len(c) for c in bad]),
                                          0, raw, hc, hu, xy, (C.c_char_p * singles)(*single_src), b"none") == -1


@pytest.mark.parametrize("compression", ["zstd", "gzip", "none", "snappy", "lz4"])
def test_storage_output_writes_the_19_column_parquet_record(host, vko, tmp_path, compression):
    """StorageOutput: files readable by pyarrow, schema = src/storage/schema.rs:78-111 field for field,
    values = the oracle's DerivedKey, partition layout and rotation = src/storage/parquet_backend.rs:86-131."""
    import datetime
    import glob
    pa = pytest.importorskip("pyarrow")
    pq = pytest.importorskip("pyarrow.parquet")
    rnd = random.Random(33)
    n = 2500
    keys = [bytes(31) + b"\x01", bytes.fromhex("c4bbcb1fbec99d65bf59d85c8cb62ee2db963f0fe106f483d9afa73bd4e39a8a")] + [rnd.getrandbits(rnd.choice([8, 64, 200, 256])).to_bytes(32, "big") for _ in range(n - 2)]
    ds = [vko.derive(k) for k in keys]
    raw = b"".join(bytes(d.raw) for d in ds)
    hc, hu = b"".join(bytes(d.h160_c) for d in ds), b"".join(bytes(d.h160_u) for d in ds)
    xy = b"".join(bytes(d.pub65)[1:] for d in ds)
    sources = [f"src{i}[x:{i % 7}]".encode() for i in range(n)]
    matched = [ds[i].p2pkh_compressed if i % 500 == 3 else b"" for i in range(n)]
    host.vkh_storage_write.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                       C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_char_p]
    nfiles = host.vkh_storage_write(str(tmp_path).encode(), b"sha256_chain:iterated", 1000, n, raw, hc, hu, xy, (C.c_char_p * n)(*sources), (C.c_char_p * n)(*matched),
                                    compression.encode())
    assert nfiles == 3                                           # 1000 + 1000 + 500 records
    assert host.vkh_storage_write(str(tmp_path / "x").encode(), b"t", 0, 0, raw, hc, hu, xy, (C.c_char_p * n)(*sources), (C.c_char_p * n)(*matched), b"brotli") == -1
    today = datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d")
    files = sorted(glob.glob(str(tmp_path / "transform=sha256_chain:iterated" / f"date={today}" / "chunk_*.parquet")))
    assert [os.path.basename(f) for f in files] == ["chunk_0001.parquet", "chunk_0002.parquet", "chunk_0003.parquet"]
    tables = [pq.read_table(f) for f in files]
    assert [t.num_rows for t in tables] == [1000, 1000, 500]
    # codec id 5 (Hadoop-framed LZ4): pyarrow decodes it but its Python metadata wrapper has no name for it
    assert pq.read_metadata(files[0]).row_group(0).column(8).compression in {"zstd": ["ZSTD"], "gzip": ["GZIP"], "none": ["UNCOMPRESSED"], "snappy": ["SNAPPY"],
                                                                              "lz4": ["LZ4", "LZ4_HADOOP", "UNKNOWN"]}[compression]
    if compression in ("snappy", "lz4"):  # the matchers written here do compress: the binary-digit column is 256 characters of '0' / '1' per record
        m8 = pq.read_metadata(files[0]).row_group(0).column(8)
        assert m8.total_compressed_size < m8.total_uncompressed_size // 2
    t = pa.concat_tables(tables)
    utf8 = lambda name, nullable: pa.field(name, pa.string(), nullable=nullable)
    want_schema = pa.schema([
        utf8("source", False), utf8("transform", False), utf8("chain", False), pa.field("timestamp", pa.timestamp("ms", tz="UTC"), nullable=False),
        utf8("matched_target", True), pa.field("private_key_raw", pa.binary(32), nullable=False), utf8("private_key_hex", False),
        utf8("private_key_decimal", False), utf8("private_key_binary", False), pa.field("private_key_bit_length", pa.uint16(), nullable=False),
        pa.field("private_key_hamming_weight", pa.uint16(), nullable=False), pa.field("private_key_leading_zeros", pa.uint8(), nullable=False),
        utf8("pubkey_compressed", True), utf8("pubkey_uncompressed", True), utf8("address_p2pkh_compressed", True),
        utf8("address_p2pkh_uncompressed", True), utf8("address_p2wpkh", True), utf8("wif_compressed", True), utf8("wif_uncompressed", True)])
    assert t.schema.equals(want_schema), t.schema
    col = lambda name: t.column(name).to_pylist()
    assert col("source") == [s.decode() for s in sources] and set(col("transform")) == {"sha256_chain:iterated"} and set(col("chain")) == {"bitcoin"}
    assert col("matched_target") == [m.decode() if m else None for m in matched]
    assert col("private_key_raw") == [bytes(d.raw) for d in ds]
    for name, get in (("private_key_hex", lambda d: d.private_key_hex), ("private_key_decimal", lambda d: d.private_key_decimal),
                      ("private_key_binary", lambda d: d.private_key_binary), ("address_p2pkh_compressed", lambda d: d.p2pkh_compressed),
                      ("address_p2pkh_uncompressed", lambda d: d.p2pkh_uncompressed), ("address_p2wpkh", lambda d: d.p2wpkh),
                      ("wif_compressed", lambda d: d.wif_compressed), ("wif_uncompressed", lambda d: d.wif_uncompressed)):
        assert col(name) == [get(d).decode() for d in ds], name
    assert col("pubkey_compressed") == [bytes(d.pub33).hex() for d in ds] and col("pubkey_uncompressed") == [bytes(d.pub65).hex() for d in ds]
    assert col("private_key_bit_length") == [d.bit_length for d in ds] and col("private_key_hamming_weight") == [d.hamming_weight for d in ds]
    assert col("private_key_leading_zeros") == [d.leading_zeros for d in ds]
    now = datetime.datetime.now(datetime.timezone.utc)
    assert all(abs((now - ts).total_seconds()) < 600 for ts in col("timestamp")[::250])


def test_bitimage_host_pieces(host):
    """The host half of Bitimage (src/bitimage.rs:38-44, 77-117, 277-293): streamed base64 + SHA-256 of file bytes against
    hashlib/base64, parse_derivation_path and increment_path_index against the reference's test cases and, on generated
    path texts, against the oracle's restatement (same accept / reject decision, same components, same error class)."""
    import base64
    import hashlib
    import random
    from oracle import vko
    host.vkh_bitimage_entropy.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    host.vkh_bitimage_entropy.restype = None
    rnd = random.Random(77)
    for n in list(range(0, 200)) + [255, 256, 257, 4095, 4096, 4097, 65536, 100003]:
        data = rnd.randbytes(n)
        out = C.create_string_buffer(32)
        host.vkh_bitimage_entropy(data, n, out)
        assert out.raw == hashlib.sha256(base64.b64encode(data)).digest(), n
    host.vkh_parse_derivation_path.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.c_int, C.c_char_p, C.c_size_t]
    host.vkh_increment_path_index.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    host.vkh_increment_path_index.restype = C.c_size_t

    def parse_path(p):
        comps, err = (C.c_uint32 * 64)(), C.create_string_buffer(256)
        d = host.vkh_parse_derivation_path(p.encode(), comps, 64, err, 256)
        return [comps[i] for i in range(d)] if d >= 0 else err.value.decode()

    def inc(p):
        o = C.create_string_buffer(600)
        n = host.vkh_increment_path_index(p.encode(), o, 600)
        return o.raw[:n].decode()

    H = 0x80000000
    assert parse_path("m/84'/0'/0'/0/0") == [84 | H, H, H, 0, 0]
    assert parse_path("m") == []
    assert parse_path("84'/0'/0'/0/0") == "Invalid derivation path: Path must start with 'm/' or be 'm'"
    assert parse_path("m/abc") == "Invalid derivation path: Invalid index: abc"
    assert parse_path("m/2147483648'/0'/0'/0/0") == "Invalid derivation path: Index 2147483648 exceeds BIP32 limit (must be < 2^31)"
    assert parse_path("m/2147483647'/0'/0'/0/0") == [0xFFFFFFFF, H, H, 0, 0]
    assert inc("m/84'/0'/0'/0/0") == "m/84'/0'/0'/0/1" and inc("m/84'/0'/0'/0/5") == "m/84'/0'/0'/0/6" and inc("m/44'/0'/0'/0'") == "m/44'/0'/0'/1'"
    pieces = ["m", "M", "", "/", "0", "1", "7", "84", "2147483647", "2147483648", "4294967295", "4294967296", "'", "h", "+", "-", " ", "\t", "x", "00"]
    for _ in range(3000):
        p = "".join(rnd.choice(pieces) for _ in range(rnd.randrange(1, 9)))
        if rnd.random() < 0.6:
            p = "m/" + p
        want = vko.bip32_parse_path(p)
        got = parse_path(p)
        if isinstance(want, int):
            assert isinstance(got, str) and {-1: "must start", -2: "Invalid index", -3: "exceeds BIP32"}[want] in got, (p, want, got)
        else:
            assert got == want, p
        assert inc(p) == vko.bip32_increment_path(p), p


def test_page_compressors_round_trip(host):
    """The Snappy and LZ4 matchers written for the Parquet writer, decoded by pyarrow's codecs: empty and tiny inputs (the
    LZ4 end-of-block rules: last 5 bytes literal, last match 12 bytes before the end), long runs, overlapping copies,
    incompressible data, literals longer than 60 / 15, matches longer than 64 / 19, distances at the 64 KB window edge."""
    pa = pytest.importorskip("pyarrow")
    host.vkh_compress_page.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    host.vkh_compress_page.restype = C.c_long
    rnd = random.Random(91)
    cases = [bytes(n) for n in range(0, 40)] + [rnd.randbytes(n) for n in list(range(0, 40)) + [59, 60, 61, 255, 256, 257, 65535, 65536, 65537, 200000]]
    cases += [b"ab" * n for n in (1, 2, 5, 6, 7, 8, 9, 30, 31, 32, 33, 34, 1000, 40000)]
    cases += [b"x" * 300000, rnd.randbytes(100) * 3000, bytes(rnd.choice(b"01") for _ in range(70000))]
    blk = rnd.randbytes(64)
    cases += [blk + rnd.randbytes(65535 - 64) + blk + b"tail-bytes-here", blk + rnd.randbytes(65536 - 64) + blk + b"tail-bytes-here"]
    cases += [b"".join(rnd.choice([b"0", b"1", b"src", b"[x:", b"]"]) for _ in range(50000))]
    for data in cases:
        out = C.create_string_buffer(len(data) * 2 + 64)
        n = host.vkh_compress_page(1, data, len(data), out, len(out))
        assert n > 0
        assert pa.Codec("snappy").decompress(out.raw[:n], decompressed_size=len(data)).to_pybytes() == data, len(data)
        n = host.vkh_compress_page(5, data, len(data), out, len(out))
        assert n >= 9 and int.from_bytes(out.raw[:4], "big") == len(data) and int.from_bytes(out.raw[4:8], "big") == n - 8
        if data:
            assert pa.Codec("lz4_raw").decompress(out.raw[8:n], decompressed_size=len(data)).to_pybytes() == data, len(data)
        if len(data) >= 1000 and len(set(data)) <= 8:
            assert n < len(data) * 2 // 3   # random binary digits: one bit of information per byte, matches of 4 and more only


def test_line_reader_follows_rust_lines_and_trim(host, tmp_path):
    """BufRead::lines() + filter_map(Result::ok) drops a line that is not UTF-8, str::trim() strips Unicode
    White_Space (src/source/wordlist.rs:24-30); Matcher::load propagates the UTF-8 error (src/matcher.rs:50-56)."""
    out = C.create_string_buffer(256)

    def f(raw):
        n = host.vkh_line_filter(raw, len(raw), out, 256)
        return None if n == -2 else out.raw[:n]

    assert f(b"  hello \t\r") == b"hello"
    assert f("\u00a0pass\u3000word\u2003\u0085".encode()) == "pass\u3000word".encode()      # NBSP, ideographic space, em space, NEL
    assert f("\u2028\u2029\u202f\u205f\u1680".encode()) == b""
    assert f("caf\u00e9 ".encode()) == "caf\u00e9".encode()
    assert f(b"caf\xe9") is None and f(b"\xc0\xaf") is None and f(b"\xed\xa0\x80") is None and f(b"\xf4\x90\x80\x80") is None  # latin-1, overlong, surrogate, > U+10FFFF
    assert f("\u200bzero-width is not White_Space\u200b".encode()) == "\u200bzero-width is not White_Space\u200b".encode()
    for s in ("", "a", "\u00e9\u00a0", "\U0001F600 x \u2009"):
        assert f(s.encode()) == s.strip(" \t\n\r\x0b\x0c\x85\xa0\u1680\u2000\u2001\u2002\u2003\u2004\u2005\u2006\u2007\u2008\u2009\u200a\u2028\u2029\u202f\u205f\u3000").encode()
    # Matcher::load: NBSP-padded address is trimmed, a latin-1 line is an error
    t = tmp_path / "t.txt"
    t.write_bytes("\u00a01BgGZ9tcN4rm9KBzDn7KprQz87SZ26SAMH\u00a0\n# c\n".encode())
    h160, kind, n = C.create_string_buffer(20 * 4), C.create_string_buffer(4), C.c_size_t()
    assert host.vkh_matcher_load(str(t).encode(), h160, kind, 4, C.byref(n)) == 1 and n.value == 1
    t.write_bytes(b"1BgGZ9tcN4rm9KBzDn7KprQz87SZ26SAMH\ncaf\xe9\n")
    assert host.vkh_matcher_load(str(t).encode(), h160, kind, 4, C.byref(n)) == -1


def test_line_reader_batches_respect_the_device_bounds_and_the_references_cut_off(host, tmp_path):
    """the stdin source (strict) hands out a group of 1000 lines only once the whole group is read and valid -- in as many
    device batches as max_lines / max_bytes ask for, never more lines or bytes than that whatever the lines' lengths -- and a line
    that is not UTF-8 loses exactly the lines of its own group (src/source/stdin.rs:45-60); the wordlist source drops the line
    (src/source/wordlist.rs:33-36).  Until round 4 a strict batch grew to the next multiple of 1000 lines past both limits."""
    host.vkh_line_batches.argtypes = [C.c_char_p, C.c_int, C.c_size_t, C.c_size_t, C.c_char_p, C.c_char_p, C.c_size_t]
    rnd = random.Random(11)

    def run(blob, strict, max_lines, max_bytes):
        src, dst = tmp_path / "in.txt", tmp_path / "out.txt"
        src.write_bytes(blob)
        err = C.create_string_buffer(256)
        rc = host.vkh_line_batches(str(src).encode(), strict, max_lines, max_bytes, str(dst).encode(), err, 256)
        batches, cur = [], []
        for ln in dst.read_bytes().split(b"\n")[:-1]:
            if ln == b"--":
                batches.append(cur)
                cur = []
            else:
                cur.append(ln)
        assert cur == []
        return rc, err.value.decode(), batches

    def lines(n, bad_at=None, longest=40):
        out = [b"  w%06d-%s \t" % (i, bytes(rnd.choice(b"abcdefgh") for _ in range(rnd.randrange(0, longest)))) for i in range(n)]
        if bad_at is not None:
            out[bad_at] = b"caf\xe9"
        return out

    for n, bad_at, max_lines, max_bytes in ((3500, None, 1000, 1 << 20), (3500, 2700, 1000, 1 << 20), (3500, 2700, 300, 1 << 20), (3500, 2700, 7, 1 << 20),
                                            (3500, 999, 64, 1 << 20), (3500, 1000, 64, 1 << 20), (2500, 2499, 100000, 4096), (2500, None, 100000, 100),
                                            (1700, 1650, 100000, 1 << 30), (5, None, 2, 50), (0, None, 10, 10)):
        raw = lines(n, bad_at)
        blob = b"\n".join(raw) + (b"\n" if raw else b"") + b"\n\n   \n"       # empty / blank lines are skipped by both sources
        good = [ln.strip() for ln in raw]
        # strict (stdin): whole groups of 1000 before the invalid line, the error afterwards
        rc, err, batches = run(blob, 1, max_lines, max_bytes)
        want = good if bad_at is None else good[:bad_at - bad_at % 1000]
        assert [ln for b in batches for ln in b] == want, (n, bad_at, max_lines, max_bytes)
        assert (rc, err) == ((0, "") if bad_at is None else (1, "stream did not contain valid UTF-8"))
        for b in batches:
            assert 0 < len(b) <= max_lines and (len(b) == 1 or sum(map(len, b)) <= max_bytes)
        # the wordlist source: the invalid line is dropped, nothing else
        rc, err, batches = run(blob, 0, max_lines, max_bytes)
        assert [ln for b in batches for ln in b] == [ln for i, ln in enumerate(good) if i != bad_at] and rc == 0
        for b in batches:
            assert 0 < len(b) <= max_lines and (len(b) == 1 or sum(map(len, b)) <= max_bytes)
    # a line of many bytes inside a group: the group is cut around it, no batch but its own exceeds the byte limit
    raw = [b"a" * 10] * 400 + [b"B" * 50000] + [b"c" * 10] * 800
    rc, err, batches = run(b"\n".join(raw) + b"\n", 1, 512, 4096)
    assert rc == 0 and [ln for b in batches for ln in b] == raw and max(len(b) for b in batches) <= 512
    assert [len(b) for b in batches if sum(map(len, b)) > 4096] == [1]
    # no final newline, CRLF line ends (trimmed like any white space)
    rc, err, batches = run(b"one\r\ntwo\r\nthree", 1, 10, 100)
    assert rc == 0 and batches == [[b"one", b"two", b"three"]]
    # lines that straddle the reader's 8 MB reads
    raw = [b"%07d" % i + b"z" * 990 for i in range(10500)]
    for strict in (0, 1):
        rc, err, batches = run(b"\n".join(raw) + b"\n", strict, 4000, 1 << 30)
        assert rc == 0 and [ln for b in batches for ln in b] == raw and max(len(b) for b in batches) <= 4000
