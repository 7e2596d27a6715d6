"""Full-size parity properties on the GPU (sizes the CPU oracle cannot sweep): the whole 2^32
Milksad space with planted targets, digest invariance under partitioning and under a different
table geometry, and the LCG per-variant key counts across the 2^31 / 2^32 seed ceilings."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import vuke_amd
    e = vuke_amd.Engine(0)
    yield e
    e.close()


def fold(parts):
    d = [0, 0, 0, 0]
    for p in parts:
        for i in range(3):
            d[i] ^= p[i]
        d[3] = (d[3] + p[3]) & (2**64 - 1)
    return tuple(d)


def test_full_2p32_milksad_sweep_finds_every_planted_seed(eng):
    from vuke_amd import _ffi as ffi
    spec = ffi.spec(ffi.T_MILKSAD)
    # planted seeds, private keys and hash160s are the ORACLE's (tests/golden/milksad_planted.json), not the device's own
    import bench_workloads as wl
    pl = wl.load_golden("milksad_planted.json")["planted"]
    planted = [p["seed"] for p in pl]
    assert planted == wl.c2_planted_seeds()
    keys = [bytes.fromhex(p["privkey"]) for p in pl]
    h160c = [bytes.fromhex(p["h160_c"]) for p in pl]
    eng.targets_set(h160c, [ffi.TARGET_P2PKH] * len(h160c))
    st, hits = eng.sweep(spec, 0, 2**32 - 1, flags=ffi.F_COMPRESSED, hit_cap=4096)
    eng.targets_set([], [])
    assert (st["inputs"], st["keys"], st["matches"], st["hits_dropped"]) == (2**32, 2**32, len(planted), 0)
    assert [h["seed"] for h in hits] == planted
    assert all(h["addr_kind"] == 0 and h["h160"] == c and h["privkey"] == k for h, c, k in zip(hits, h160c, keys))
    g = wl.load_golden("milksad_2p32.json")  # the oracle's digest of the whole space (default W=24 engine here)
    assert g is not None and wl.hexd(st["digest"]) == g["digest"]
    # the same space in 16 blocks: counters add up, digest folds to the same value
    parts = [eng.sweep(spec, b << 28, ((b + 1) << 28) - 1, flags=ffi.F_COMPRESSED)[0] for b in range(16)]
    assert sum(p["keys"] for p in parts) == 2**32
    assert fold(p["digest"] for p in parts) == st["digest"]
    # seeds past u32::MAX are skipped by the transform (src/transform/milksad.rs:21): no keys, empty digest
    st2, _ = eng.sweep(spec, 2**32, 2**32 + 2**24, flags=ffi.F_COMPRESSED)
    assert st2["keys"] == 0 and st2["digest"] == (0, 0, 0, 0)


def test_bench_geometry_full_2p32_sweep_against_the_oracles_golden():
    """BASELINE config 2 end to end on the geometry the headline is quoted on (bench.py: layout 1 = signed9 / 120 GB,
    batch 256, 2 workgroups per CU): all 16 blocks of 2^28 Milksad seeds, every expected value from the CPU oracle --
    the planted seeds / keys / hash160s (milksad_planted.json) and the digest of each 2^28 block and of the whole 2^32
    space (milksad_2p32.json, tests/golden/make_milksad_2p32.py: the oracle over all 2^32 seeds).
    Reference behaviour: src/transform/milksad.rs:17-29, src/derive.rs:78-118, src/matcher.rs:72-98."""
    import vuke_amd
    from vuke_amd import _ffi as ffi
    import bench_workloads as wl
    g = wl.load_golden("milksad_2p32.json")
    assert g is not None, "tests/golden/milksad_2p32.json is missing: run tests/golden/make_milksad_2p32.py"
    pl = wl.load_golden("milksad_planted.json")["planted"]
    e = vuke_amd.Engine(0, layout=1, batch=256)
    try:
        info = e.info()
        assert (info["layout"], info["windows"], info["batch"]) == (1, 9, 256)
        spec = ffi.spec(ffi.T_MILKSAD)
        e.targets_set([bytes.fromhex(p["h160_c"]) for p in pl], [ffi.TARGET_P2PKH] * len(pl))
        parts, hits = [], []
        for b in range(16):
            st, h = e.sweep(spec, b << 28, ((b + 1) << 28) - 1, flags=ffi.F_COMPRESSED, hit_cap=2048)
            assert (st["inputs"], st["keys"], st["hits_dropped"]) == (2**28, 2**28, 0)
            assert wl.hexd(st["digest"]) == g["block_digests"][b], f"block {b}"
            parts.append(st["digest"])
            hits += h
        assert wl.hexd(wl.fold_digests(parts)) == g["digest"]
        assert [(h["seed"], h["privkey"].hex(), h["h160"].hex(), h["addr_kind"]) for h in hits] == [(p["seed"], p["privkey"], p["h160_c"], 0) for p in pl]
        # a few of the 256 oracle windows of 2^24 seeds on their own, and one full-range call (4 launches of 2^30 inside the library)
        for w in (0, 37, 255):
            st, _ = e.sweep(spec, w << 24, ((w + 1) << 24) - 1, flags=ffi.F_COMPRESSED)
            assert wl.hexd(st["digest"]) == g["window_digests"][w], f"window {w}"
        st, h = e.sweep(spec, 0, 2**32 - 1, flags=ffi.F_COMPRESSED, hit_cap=2048)
        assert (st["keys"], st["matches"], wl.hexd(st["digest"])) == (2**32, len(pl), g["digest"])
    finally:
        e.close()


def test_digest_independent_of_table_geometry_and_batch(eng):
    import vuke_amd
    from vuke_amd import _ffi as ffi
    spec = ffi.spec(ffi.T_SHA256)
    lo, hi = 10**9, 10**9 + 2**22 - 1
    ref, _ = eng.sweep(spec, lo, hi, flags=3)
    assert ref["keys"] == 3 * 2**22
    for W, batch, bpc in ((16, 8, 1), (19, 48, 3)):
        e = vuke_amd.Engine(0, window_bits=W, batch=batch, blocks_per_cu=bpc)
        st, _ = e.sweep(spec, lo, hi, flags=3)
        e.close()
        assert st["digest"] == ref["digest"] and st["keys"] == ref["keys"]
    # split at an arbitrary point
    a, _ = eng.sweep(spec, lo, lo + 1234567, flags=3)
    b, _ = eng.sweep(spec, lo + 1234568, hi, flags=3)
    assert fold([a["digest"], b["digest"]]) == ref["digest"]


def test_lcg_all_variants_key_counts_across_ceilings(eng):
    from vuke_amd import _ffi as ffi
    spec = ffi.spec(ffi.T_LCG)
    h = 2**22
    # around 2^31: glibc valid < 2^31, minstd valid < 2^31 - 1, msvc/borland valid <= u32::MAX
    st, _ = eng.sweep(spec, 2**31 - h, 2**31 + h - 1, flags=1)
    assert st["inputs"] == 2 * h and st["keys"] == h + (h - 1) + 2 * h + 2 * h
    st, _ = eng.sweep(spec, 2**32 - h, 2**32 + h - 1, flags=1)
    assert st["keys"] == 2 * h


def _hit_records(hits):
    return sorted((h["seed"], h["sub_index"], h["addr_kind"], h["privkey"].hex(), h["h160"].hex()) for h in hits)


def _planted_records(planted):
    return sorted((p["seed"], p["sub_index"], p["addr_kind"], p["privkey"], p["h160"]) for p in planted)


def test_config3_sha256_scan_against_a_million_targets(eng):
    """BASELINE config 3 at full size: seeds 0..99 999 999 x 3 keys, 1 000 000 targets = 999 000 decoys +
    1 000 planted (seed 7919*j, slot j%3, kind cycling p2pkh_c / p2pkh_u / p2wpkh).  The planted hash160s and the expected
    hit records (seed, slot, address kind, private key, hash160) are the ORACLE's (tests/golden/bench_c3.json,
    make_bench_fixtures.py) -- nothing the device computed is used to build the expectation."""
    import bench_workloads as wl
    from vuke_amd import _ffi as ffi
    spec = ffi.spec(ffi.T_SHA256)
    fx = wl.load_golden("bench_c3.json")
    planted = fx["planted"]
    assert [(p["seed"], p["sub_index"], p["addr_kind"]) for p in planted] == wl.c3_planted()
    h160s = [bytes.fromhex(p["h160"]) for p in planted]
    kinds = [ffi.TARGET_P2WPKH if p["addr_kind"] == 2 else ffi.TARGET_P2PKH for p in planted]
    for d, kind in wl.c3_decoys():
        h160s.append(d)
        kinds.append(kind)
    eng.targets_set(h160s, kinds)
    st, hits = eng.sweep(spec, 0, wl.C3_INPUTS - 1, flags=3, hit_cap=4096)
    eng.targets_set([], [])
    assert (st["inputs"], st["keys"], st["matches"], st["hits_dropped"]) == (10**8, 3 * 10**8, 1000, 0)
    assert _hit_records(hits) == _planted_records(planted)


def test_config4_all_lcg_variants_over_2p31_seeds(eng):
    """BASELINE config 4 at full size on one GPU: every LCG variant x seeds 0..2^31-1 in eight contiguous 2^28 blocks
    (the 8-GPU split), 256 planted keys per variant with hash160 and private key from the ORACLE (tests/golden/bench_c4.json);
    counters, full hit records and the digest of the whole agree with the blocks."""
    import bench_workloads as wl
    from vuke_amd import _ffi as ffi
    spec = ffi.spec(ffi.T_LCG)
    planted = wl.load_golden("bench_c4.json")["planted"]
    assert [(p["seed"], p["sub_index"]) for p in planted] == wl.c4_planted() and len(planted) == 1024
    eng.targets_set([bytes.fromhex(p["h160"]) for p in planted], [ffi.TARGET_P2PKH] * len(planted))
    parts, hits = [], []
    for b in range(8):
        st, h = eng.sweep(spec, b << 28, ((b + 1) << 28) - 1, flags=ffi.F_COMPRESSED, hit_cap=4096)
        parts.append(st)
        hits += h
    eng.targets_set([], [])
    assert sum(p["inputs"] for p in parts) == 2**31
    assert sum(p["keys"] for p in parts) == 4 * 2**31 - 1          # minstd skips seed 2^31 - 1 (src/transform/lcg.rs: seed <= m - 1)
    assert sum(p["matches"] for p in parts) == len(planted) and sum(p["hits_dropped"] for p in parts) == 0
    assert _hit_records(hits) == _planted_records(planted)
    whole, _ = eng.sweep(spec, 0, 2**31 - 1, flags=ffi.F_COMPRESSED)
    assert whole["keys"] == 4 * 2**31 - 1 and whole["digest"] == fold(p["digest"] for p in parts)
    assert wl.hexd(whole["digest"]) == wl.load_golden("bench_c4_full.json")["digest"]   # the oracle's, over the whole config


def test_config5_key_chains_at_size(eng):
    """BASELINE config 5 at its stated size, BOTH chains over all 100 000 inputs (the bench workload: bench_workloads.c5_mnemonics /
    c5_electrum_seeds), through vk_scan_strings.  Everything expected is the ORACLE's: the planted hash160s and private keys (key 7
    of every 1000th input, tests/golden/bench_c5.json -- MultiBit planted on the compressed form, Electrum on the uncompressed one,
    src/electrum.rs:216-285) and counters + digest over the 2 000 000 keys of each chain (tests/golden/bench_c5_full.json,
    make_c5_full.py).  Whole hit records are compared, as for configs 3 / 4 (rounds 2-3 planted hash160s the
    device itself had derived and ran a 4 096-seed slice of Electrum).  Reference: src/transform/multibit.rs:45-61,
    src/transform/electrum.rs:43-67."""
    import bench_workloads as wl
    from vuke_amd import _ffi as ffi
    fx, full = wl.load_golden("bench_c5.json"), wl.load_golden("bench_c5_full.json")
    eng.bip39_wordlist_set(wl.c5_wordlist())
    for which, spec, lines in (("multibit", ffi.spec(ffi.T_MULTIBIT, 0, 0, 20), wl.c5_mnemonics()),
                               ("electrum", ffi.spec(ffi.T_ELECTRUM, 0, 0, 20), wl.c5_electrum_seeds())):
        planted = fx[which]["planted"]
        assert len(lines) == wl.C5_INPUTS == 100_000 and lines[0].decode() == fx[which]["first_input"]
        assert [p["input"] for p in planted] == list(range(0, 100_000, 1000)) and all(p["sub_index"] == 7 for p in planted)
        eng.targets_set([bytes.fromhex(p["h160"]) for p in planted], [ffi.TARGET_P2PKH] * len(planted))
        st, hits = eng.scan_strings(spec, lines, flags=3, hit_cap=1024)
        eng.targets_set([], [])
        g = full[which]
        assert (st["inputs"], st["keys"], st["matches"], st["hits_dropped"], st["inputs_skipped"]) == (g["inputs"], g["keys"], len(planted), 0, 0)
        assert g["keys"] == 2_000_000 and g["flags"] == 3 and wl.hexd(st["digest"]) == g["digest"], which
        got = sorted((h["index"], h["sub_index"], h["addr_kind"], h["privkey"].hex(), h["h160"].hex()) for h in hits)
        assert got == sorted((p["input"], p["sub_index"], p["addr_kind"], p["privkey"], p["h160"]) for p in planted), which


def test_configs_3_and_4_at_their_stated_sizes_against_the_oracle(eng):
    """BASELINE config 3 (sha256 transform, 3 keys per input, both hash160s, inputs 0 .. 3*2^25-1 >= the stated 10^8) and
    config 4 (all four LCG variants x the WHOLE 2^31 seed space, 4*2^31 - 1 keys) block by block against the CPU oracle's
    counters and digests (tests/golden/bench_c3_full.json, bench_c4_full.json; make_c34_full.py, ~9 x 10^9 oracle keys)."""
    from vuke_amd import _ffi as ffi
    import bench_workloads as wl
    for name, spec in (("bench_c3_full.json", ffi.spec(ffi.T_SHA256)), ("bench_c4_full.json", ffi.spec(ffi.T_LCG))):
        g = wl.load_golden(name)
        assert g is not None, f"tests/golden/{name} is missing: run tests/golden/make_c34_full.py"
        parts, keys = [], 0
        for b in g["blocks"]:
            st, _ = eng.sweep(spec, b["lo"], b["hi"], flags=g["flags"])
            assert (st["inputs"], st["keys"], wl.hexd(st["digest"])) == (b["inputs"], b["keys"], b["digest"]), (name, b["lo"])
            parts.append(st["digest"])
            keys += st["keys"]
        assert keys == g["keys"] and wl.hexd(wl.fold_digests(parts)) == g["digest"]
    assert wl.load_golden("bench_c4_full.json")["keys"] == 4 * 2**31 - 1   # minstd skips seed 2^31 - 1 (src/lcg.rs:77-83)


def test_randomized_parity_on_the_bench_geometry(vko):
    """the geometry the headline is quoted on (layout 1 = signed9 / 120 GB, batch 256, 2 workgroups per
    CU) against the ORACLE in a driver-run test: random transform specs, ranges around every seed ceiling, flags -- the case
    generator of tools/fuzz_parity.py -- plus a 2^22-seed Milksad soak.  Counters and digest must equal the oracle's."""
    import os
    import random
    import vuke_amd
    from vuke_amd import _ffi as ffi
    rnd = random.Random(20261004)
    threads = min(16, len(os.sched_getaffinity(0)))
    e = vuke_amd.Engine(0, layout=1, batch=256)
    try:
        edges = [0, 2**31 - 1, 2**31, 2**32 - 1, 2**32, 2**63, 2**64 - 1, 10**9, 10**18]
        for c in range(160):
            kind = rnd.choice([ffi.T_SHA256, ffi.T_MILKSAD, ffi.T_MT64, ffi.T_LCG, ffi.T_XORSHIFT, ffi.T_SHA256_CHAIN, ffi.T_DOUBLE_SHA256, ffi.T_MD5, ffi.T_DIRECT])
            variant = rnd.choice([ffi.VARIANT_ALL, 0, 1, 2, 3]) if kind in (ffi.T_LCG, ffi.T_XORSHIFT, ffi.T_SHA256_CHAIN) else ffi.VARIANT_ALL
            endian = rnd.randrange(2) if kind == ffi.T_LCG else 0
            depth = rnd.choice([1, 2, 7, 10]) if kind == ffi.T_SHA256_CHAIN else 10
            gs, osp = ffi.spec(kind, variant, endian, depth), vko.spec(kind, variant, endian, depth)
            slots = vko.slots_per_input(osp)
            n = rnd.choice([1, 2, 255, 256, 257, 1000, 65536, rnd.randrange(1, 1 << 17)])
            n = max(1, min(n, (1 << 17) // slots))
            centre = rnd.choice(edges + [rnd.getrandbits(rnd.randrange(1, 65))])
            lo = max(0, min(centre - rnd.randrange(0, n + 1), 2**64 - n))
            hi = lo + n - 1
            flags = rnd.choice([1, 2, 3])
            st, _ = e.sweep(gs, lo, hi, flags=flags)
            ost, _ = vko.sweep(osp, lo, hi, flags=flags | vko.F_BATCHED, nthreads=threads)
            assert (st["inputs"], st["keys"], st["digest"]) == (ost["inputs"], ost["keys"], ost["digest"]), (c, kind, variant, endian, depth, lo, hi, flags)
        lo = 3_000_000_000
        st, _ = e.sweep(ffi.spec(ffi.T_MILKSAD), lo, lo + 2**22 - 1, flags=3)
        ost, _ = vko.sweep(vko.spec(vko.T_MILKSAD), lo, lo + 2**22 - 1, flags=3 | vko.F_BATCHED, nthreads=threads)
        assert (st["keys"], st["digest"]) == (ost["keys"], ost["digest"]) and st["keys"] == 2**22
    finally:
        e.close()


def test_large_oracle_windows_of_every_other_transform(eng):
    """One large oracle-swept window (16.8 M - 67 M keys, both hash160s) per transform kind that has no full-size golden of
    its own: mt64, xorshift x4, sha256_chain (80 keys per input), double_sha256, md5, direct, lcg little-endian, armory --
    around the seed ceilings where there is one (tests/golden/soak_windows.json, make_soak_windows.py: 3.7 x 10^8 oracle keys)."""
    import bench_workloads as wl
    from vuke_amd import _ffi as ffi
    g = wl.load_golden("soak_windows.json")
    assert g is not None and len(g["windows"]) == 8
    for w in g["windows"]:
        st, _ = eng.sweep(ffi.spec(*w["spec"]), w["lo"], w["hi"], flags=w["flags"])
        assert (st["inputs"], st["keys"], wl.hexd(st["digest"])) == (w["inputs"], w["keys"], w["digest"]), w["name"]
