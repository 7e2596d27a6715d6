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
    planted = sorted({(0x9E3779B1 * j + 12345) & 0xFFFFFFFF for j in range(1024)} | {0, 2**32 - 1})
    keys = [row[0] for row in eng.transform_batch(spec, planted)]
    _, h160c, _ = eng.derive_batch(keys)
    eng.targets_set(h160c, [ffi.TARGET_P2PKH] * len(h160c))
    st, hits = eng.sweep(spec, 0, 2**32 - 1, flags=ffi.F_COMPRESSED, hit_cap=4096)
    eng.targets_set([], [])
    assert (st["inputs"], st["keys"], st["matches"], st["hits_dropped"]) == (2**32, 2**32, len(planted), 0)
    assert [h["seed"] for h in hits] == planted
    assert all(h["addr_kind"] == 0 and h["h160"] == c and h["privkey"] == k for h, c, k in zip(hits, h160c, keys))
    # the same space in 16 blocks: counters add up, digest folds to the same value
    parts = [eng.sweep(spec, b << 28, ((b + 1) << 28) - 1, flags=ffi.F_COMPRESSED)[0] for b in range(16)]
    assert sum(p["keys"] for p in parts) == 2**32
    assert fold(p["digest"] for p in parts) == st["digest"]
    # seeds past u32::MAX are skipped by the transform (src/transform/milksad.rs:21): no keys, empty digest
    st2, _ = eng.sweep(spec, 2**32, 2**32 + 2**24, flags=ffi.F_COMPRESSED)
    assert st2["keys"] == 0 and st2["digest"] == (0, 0, 0, 0)


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
