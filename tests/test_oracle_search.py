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
