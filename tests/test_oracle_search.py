"""Pin the oracle's analyzer search (oracle/transforms.c: draws, masked / cascade predicate, vko_search) against the
reference's documented vectors (README.md:413,438-441; src/analyze/milksad.rs:477-488 mask formula) and numpy."""
import numpy as np


def masked(key, bits):
    k = int.from_bytes(key[24:], "big")
    return (k & ((1 << bits) - 1)) | (1 << (bits - 1))


def test_draws_vs_numpy_and_readme(vko):
    for seed in (0, 100, 2**32 - 1, 1610000002):
        assert b"".join(vko.milksad_draws(seed, 8)) == np.random.RandomState(seed).bytes(256)
        assert vko.milksad_draws(seed, 1)[0] == vko.milksad_key(seed)
    d = vko.milksad_draws(100, 3)   # README.md:438-441
    assert [masked(d[0], 5), masked(d[1], 10), masked(d[2], 15)] == [0x16, 0x273, 0x7a85]
    assert masked(vko.milksad_key(1610000002), 5) == 0x15    # README.md:405-413
    for seed in (0, 12345, 2**64 - 1):
        assert vko.mt64_draws(seed, 1)[0] == vko.mt64_key(seed)
    assert len(set(vko.mt64_draws(7, 8))) == 8


def test_search_modes(vko):
    sp = vko.spec(vko.T_MILKSAD)
    # cascade of README.md:424-441: the seed is 100
    assert vko.search(sp, 0, 5000, cascade=[(5, 0x16), (10, 0x273), (15, 0x7a85)])[0] == 100
    key = vko.milksad_key(4321)
    assert vko.search(sp, 4000, 5000, key=key) == (4321, [key])
    assert vko.search(sp, 0, 4320, key=key) is None
    seed, keys = vko.search(sp, 0, 100000, key=(0x15).to_bytes(32, "big"), mask_bits=5)
    assert masked(keys[0], 5) == 0x15 and all(masked(vko.milksad_key(s), 5) != 0x15 for s in range(seed))
    # lcg: src/analyze/lcg.rs tests (find known glibc / minstd seed, little endian, masked)
    for variant, le in ((0, False), (1, False), (2, True)):
        sp = vko.spec(vko.T_LCG, variant, 1 if le else 0)
        k = vko.lcg_key(12345, variant, le)
        assert vko.search(sp, 0, 20000, key=k) == (12345, [k])
    assert vko.search(vko.spec(vko.T_LCG, 0, 1), 0, 20000, key=vko.lcg_key(12345, 0, False)) is None  # wrong endian
    sp = vko.spec(vko.T_MT64)
    d = vko.mt64_draws(777, 3)
    casc = [(8, masked(d[0], 8)), (16, masked(d[1], 16)), (24, masked(d[2], 24))]
    assert vko.search(sp, 0, 2000, cascade=casc) == (777, d)


def _chain(seed, variant, depth):
    """generate_chain(seed.to_be_bytes(), variant, depth) restated with hashlib (src/sha256_chain.rs:88-145)."""
    import hashlib
    s = seed.to_bytes(4, "big")
    out = []
    for i in range(depth):
        if variant == 0:
            out.append(hashlib.sha256(out[-1] if out else s).digest())
        elif variant == 1:
            out.append(hashlib.sha256(s + i.to_bytes(4, "big")).digest())
        elif variant == 2:
            out.append(hashlib.sha256(s + i.to_bytes(4, "little")).digest())
        else:
            out.append(hashlib.sha256(s + str(i).encode()).digest())
    return out


def test_xorshift_cascade(vko):
    for variant in range(4):
        for seed in (0, 1, 77, 2**40 + 5, 2**64 - 1):
            d = vko.xorshift_draws(seed, variant, 4)
            assert d[0] == vko.xorshift_key(seed, variant) and len(set(d)) == 4
        # src/analyze/xorshift.rs tests: a cascade built from a known seed finds it
        d = vko.xorshift_draws(3000, variant, 3)
        casc = [(8, masked(d[0], 8)), (16, masked(d[1], 16)), (24, masked(d[2], 24))]
        assert vko.search(vko.spec(vko.T_XORSHIFT, variant), 0, 5000, cascade=casc) == (3000, d)
    assert vko.xorshift_draws(0, 0, 2) == vko.xorshift_draws(1, 0, 2)    # seed 0 is mapped to 1


def test_sha256_chain_search_reference_cases(vko):
    # src/analyze/sha256_chain.rs:686-810 (the reference's #[ignore]d full-range tests, on bounded ranges here)
    key = _chain(12345, 0, 5)[2]
    assert vko.search(vko.spec(vko.T_SHA256_CHAIN, 0, 0, 5), 12000, 13000, key=key, detail=True) == (12345, [key], (0, 2))
    assert vko.search(vko.spec(vko.T_SHA256_CHAIN, 0, 0, 2), 12000, 13000, key=key) is None           # depth too short
    full = _chain(42, 0, 3)[0]
    tkey = bytes(24) + masked(full, 10).to_bytes(8, "big")
    seed, keys, (v, idx) = vko.search(vko.spec(vko.T_SHA256_CHAIN, 0, 0, 3), 0, 42, key=tkey, mask_bits=10, detail=True)
    assert seed <= 42 and masked(keys[0], 10) == masked(full, 10) and keys[0] == _chain(seed, 0, 3)[idx] and v == 0
    for known, variant in ((100, 0), (500, 1), (999, 3), (0, 0), (321, 2)):
        ch = _chain(known, variant, 3)
        casc = [(5, masked(ch[0], 5)), (10, masked(ch[1], 10)), (15, masked(ch[2], 15))]
        seed, keys, (v, _) = vko.search(vko.spec(vko.T_SHA256_CHAIN, variant), 0, known, cascade=casc, detail=True)
        assert seed <= known and v == variant and keys == _chain(seed, variant, 3)
        assert [masked(k, b) for k, (b, _) in zip(keys, casc)] == [t for _, t in casc]
    # all variants: the first variant in ALL_VARIANTS order that matches is the one reported
    key = _chain(777, 2, 10)[9]
    assert vko.search(vko.spec(vko.T_SHA256_CHAIN), 700, 800, key=key, detail=True) == (777, [key], (2, 9))
