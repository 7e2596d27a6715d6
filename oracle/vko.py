"""ctypes binding of the CPU oracle (oracle/build/libvko.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (vuke_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# VKO_LIB: another build of the same checker (oracle/Makefile `asan`: build/asan/libvko.so under AddressSanitizer + UBSan, with
# python started under LD_PRELOAD=libasan -- tools/run_sanitizers.sh, tests/test_sanitizers.py)
_LIB = os.environ.get("VKO_LIB") or os.path.join(_HERE, "build", "libvko.so")

T_SHA256, T_MILKSAD, T_MT64, T_LCG, T_XORSHIFT, T_SHA256_CHAIN, T_ELECTRUM, T_MULTIBIT, T_DOUBLE_SHA256, T_MD5, T_DIRECT, T_ARMORY = range(12)
VARIANT_ALL = 0xFFFFFFFF
F_COMPRESSED, F_UNCOMPRESSED, F_FAITHFUL, F_BATCHED = 1, 2, 0x100, 0x200


def build(force=False):
    if os.environ.get("VKO_LIB"):
        return _LIB
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _LIB


class Spec(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("variant", C.c_uint32), ("endian", C.c_uint32), ("chain_depth", C.c_uint32)]


class Derived(C.Structure):
    _fields_ = [
        ("raw", C.c_uint8 * 32), ("pub33", C.c_uint8 * 33), ("pub65", C.c_uint8 * 65),
        ("h160_c", C.c_uint8 * 20), ("h160_u", C.c_uint8 * 20),
        ("private_key_hex", C.c_char * 65), ("private_key_decimal", C.c_char * 80),
        ("private_key_binary", C.c_char * 257),
        ("bit_length", C.c_uint16), ("hamming_weight", C.c_uint16), ("leading_zeros", C.c_uint8),
        ("pubkey_compressed", C.c_char * 67), ("pubkey_uncompressed", C.c_char * 131),
        ("wif_compressed", C.c_char * 60), ("wif_uncompressed", C.c_char * 60),
        ("p2pkh_compressed", C.c_char * 40), ("p2pkh_uncompressed", C.c_char * 40), ("p2wpkh", C.c_char * 50),
    ]


class Hit(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("sub_index", C.c_uint32), ("addr_kind", C.c_uint32),
                ("privkey", C.c_uint8 * 32), ("h160", C.c_uint8 * 20)]


class Stats(C.Structure):
    _fields_ = [("inputs", C.c_uint64), ("keys", C.c_uint64), ("matches", C.c_uint64), ("digest", C.c_uint64 * 4)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        u8p = C.POINTER(C.c_uint8)
        L.vko_matcher_new.restype = C.c_void_p
        L.vko_matcher_free.argtypes = [C.c_void_p]
        L.vko_matcher_add_line.argtypes = [C.c_void_p, C.c_char_p]
        L.vko_matcher_count.argtypes = [C.c_void_p]
        L.vko_matcher_count.restype = C.c_size_t
        L.vko_matcher_check.argtypes = [C.c_void_p, C.POINTER(Derived)]
        L.vko_matcher_check_h160.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.vko_transform_name.restype = C.c_char_p
        L.vko_sweep.argtypes = [C.POINTER(Spec), C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p,
                                C.POINTER(Hit), C.c_size_t, C.POINTER(Stats), C.c_int]
        L.vko_transform_key.argtypes = [C.POINTER(Spec), C.c_uint64, C.c_uint32, C.c_char_p]
        L.vko_transform_label.argtypes = [C.POINTER(Spec), C.c_uint64, C.c_uint32, C.c_char_p, C.c_size_t]
        L.vko_milksad_key.argtypes = [C.c_uint64, C.c_char_p]
        L.vko_mt64_key.argtypes = [C.c_uint64, C.c_char_p]
        L.vko_lcg_key.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_char_p]
        L.vko_xorshift_key.argtypes = [C.c_uint64, C.c_uint32, C.c_char_p]
        L.vko_sha256_chain_key.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_char_p]
        cp, sz, u32, u64 = C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint64
        for name in ("vko_sha256", "vko_ripemd160", "vko_hash160", "vko_sha512"):
            getattr(L, name).argtypes = [cp, sz, cp]
        L.vko_hmac_sha512.argtypes = [cp, sz, cp, sz, cp]
        L.vko_pbkdf2_hmac_sha512.argtypes = [cp, sz, cp, sz, u32, cp, sz]
        L.vko_slots_per_input.argtypes = [C.POINTER(Spec)]
        L.vko_slots_per_input.restype = u32
        L.vko_transform_name.argtypes = [C.POINTER(Spec)]
        L.vko_normalize_key.argtypes = [cp, cp]
        L.vko_pubkey.argtypes = [cp, cp, cp]
        L.vko_pubkey_slow.argtypes = [cp, cp, cp]
        L.vko_scalar_add.argtypes = [cp, cp, cp]
        L.vko_scalar_valid.argtypes = [cp]
        L.vko_derive.argtypes = [cp, C.POINTER(Derived)]
        L.vko_derive_h160.argtypes = [cp, cp, cp]
        L.vko_fe_mul_be.argtypes = [cp, cp, cp]
        L.vko_fe_inv_be.argtypes = [cp, cp]
        L.vko_base58check_encode.argtypes = [cp, sz, cp, sz]
        L.vko_base58check_decode.argtypes = [cp, cp, sz]
        L.vko_segwit_encode.argtypes = [cp, C.c_int, cp, sz, cp, sz]
        L.vko_segwit_decode.argtypes = [cp, cp, C.POINTER(C.c_int), cp, C.POINTER(sz)]
        L.vko_electrum_stretch.argtypes = [cp, sz, u32, cp]
        L.vko_electrum_keys.argtypes = [cp, sz, u32, u32, u32, cp, cp, cp]
        L.vko_electrum_label.argtypes = [cp, u32, u32, cp, sz]
        L.vko_multibit_buggy_mnemonic.argtypes = [C.POINTER(cp), cp, cp, sz]
        L.vko_multibit_buggy_mnemonic.restype = sz
        L.vko_multibit_keys.argtypes = [C.POINTER(cp), cp, sz, u32, cp, cp, cp, cp]
        L.vko_multibit_label.argtypes = [cp, u32, cp, sz]
        L.vko_multibit_keys_pass.argtypes = [C.POINTER(cp), cp, sz, cp, sz, u32, cp, cp, cp, cp]
        L.vko_base64.argtypes = [cp, sz, cp]
        L.vko_base64.restype = sz
        L.vko_bitimage_entropy.argtypes = [cp, sz, cp]
        L.vko_bip39_mnemonic.argtypes = [C.POINTER(cp), cp, sz, cp, sz]
        L.vko_bip39_mnemonic.restype = sz
        L.vko_bip32_parse_path.argtypes = [cp, C.POINTER(u32), C.c_int]
        L.vko_bip32_increment_path.argtypes = [cp, cp, sz]
        L.vko_bip32_increment_path.restype = sz
        L.vko_bip32_derive.argtypes = [cp, C.POINTER(u32), C.c_int, cp]
        L.vko_bitimage_keys.argtypes = [C.POINTER(cp), cp, cp, sz, cp, u32, cp, cp]
        L.vko_bitimage_keys.restype = u32
        L.vko_milksad_draws.argtypes = [u32, u32, cp]
        L.vko_mt64_draws.argtypes = [u64, u32, cp]
        L.vko_search.argtypes = [C.POINTER(Spec), u64, u64, u32, cp, u32, cp, C.POINTER(u64), C.POINTER(u64), cp, C.POINTER(u32)]
        L.vko_xorshift_draws.argtypes = [u64, u32, u32, cp]
        L.vko_slots_per_string.argtypes = [C.POINTER(Spec)]
        L.vko_slots_per_string.restype = u32
        L.vko_transform_string.argtypes = [C.POINTER(Spec), cp, sz, u32, cp]
        L.vko_md5.argtypes = [cp, sz, cp]
        L.vko_hmac_sha256.argtypes = [cp, sz, cp, sz, cp]
        L.vko_scalar_mul.argtypes = [cp, cp, cp]
        L.vko_armory_key.argtypes = [cp, cp]
        _ = (u8p, u64)
        _lib = L
    return _lib


def _out(n):
    return C.create_string_buffer(n)


def sha256(m): o = _out(32); lib().vko_sha256(m, len(m), o); return o.raw
def ripemd160(m): o = _out(20); lib().vko_ripemd160(m, len(m), o); return o.raw
def hash160(m): o = _out(20); lib().vko_hash160(m, len(m), o); return o.raw
def sha512(m): o = _out(64); lib().vko_sha512(m, len(m), o); return o.raw
def hmac_sha512(k, m): o = _out(64); lib().vko_hmac_sha512(k, len(k), m, len(m), o); return o.raw


def pbkdf2_hmac_sha512(pw, salt, iters, outlen=64):
    o = _out(outlen)
    lib().vko_pbkdf2_hmac_sha512(pw, len(pw), salt, len(salt), iters, o, outlen)
    return o.raw


def milksad_key(seed):
    o = _out(32)
    return o.raw if lib().vko_milksad_key(seed, o) else None


def mt64_key(seed):
    o = _out(32)
    lib().vko_mt64_key(seed, o)
    return o.raw


def lcg_key(seed, variant, le=False):
    o = _out(32)
    return o.raw if lib().vko_lcg_key(seed, variant, 1 if le else 0, o) else None


def xorshift_key(seed, variant):
    o = _out(32)
    return o.raw if lib().vko_xorshift_key(seed, variant, o) else None


def sha256_chain_key(seed_bytes, variant, index):
    o = _out(32)
    return o.raw if lib().vko_sha256_chain_key(seed_bytes, len(seed_bytes), variant, index, o) else None


def spec(kind, variant=VARIANT_ALL, endian=0, chain_depth=10):
    return Spec(kind, variant, endian, chain_depth)


def slots_per_input(sp):
    return lib().vko_slots_per_input(C.byref(sp))


def transform_key(sp, seed, sub):
    o = _out(32)
    return o.raw if lib().vko_transform_key(C.byref(sp), seed, sub, o) else None


def slots_per_string(sp):
    return lib().vko_slots_per_string(C.byref(sp))


def transform_string(sp, s, sub):
    """Key for (Input::from_string(s), sub) or None where the reference emits nothing."""
    if isinstance(s, str):
        s = s.encode()
    o = _out(32)
    return o.raw if lib().vko_transform_string(C.byref(sp), s, len(s), sub, o) else None


def hmac_sha256(key, msg):
    o = _out(32)
    lib().vko_hmac_sha256(key, len(key), msg, len(msg), o)
    return o.raw


def scalar_mul(a, b):
    o = _out(32)
    return o.raw if lib().vko_scalar_mul(a, b, o) else None


def armory_key(seed32):
    o = _out(32)
    return o.raw if lib().vko_armory_key(seed32, o) else None


def md5(msg):
    o = _out(16)
    lib().vko_md5(msg, len(msg), o)
    return o.raw


def transform_label(sp, seed, sub):
    o = _out(128)
    n = lib().vko_transform_label(C.byref(sp), seed, sub, o, 128)
    return o.value.decode() if n >= 0 else None


def transform_name(sp):
    return lib().vko_transform_name(C.byref(sp)).decode()


def transform_apply(sp, seed):
    """Transform::apply_batch for one numeric input: ordered [(label, key)]."""
    out = []
    for sub in range(slots_per_input(sp)):
        k = transform_key(sp, seed, sub)
        if k is not None:
            out.append((transform_label(sp, seed, sub), k))
    return out


def electrum_stretch(seed, rounds=100000):
    o = _out(32)
    lib().vko_electrum_stretch(seed, len(seed), rounds, o)
    return o.raw


def electrum_keys(seed, for_change=False, count=20, rounds=100000):
    """ElectrumTransform::apply_batch for one string input: (mpk or None, [(label, key)])."""
    keys, valid, mpk = _out(32 * count), _out(count), _out(64)
    if not lib().vko_electrum_keys(seed, len(seed), 1 if for_change else 0, count, rounds, keys, valid, mpk):
        return None, []
    out = []
    for i in range(count):
        if valid.raw[i]:
            lab = _out(160)
            lib().vko_electrum_label(seed, 1 if for_change else 0, i, lab, 160)
            out.append((lab.value.decode(), keys.raw[32 * i:32 * i + 32]))
    return mpk.raw, out


def milksad_draws(seed, n):
    o = _out(32 * n)
    lib().vko_milksad_draws(seed, n, o)
    return [o.raw[32 * i:32 * i + 32] for i in range(n)]


def mt64_draws(seed, n):
    o = _out(32 * n)
    lib().vko_mt64_draws(seed, n, o)
    return [o.raw[32 * i:32 * i + 32] for i in range(n)]


def xorshift_draws(seed, variant, n):
    o = _out(32 * n)
    lib().vko_xorshift_draws(seed, variant, n, o)
    return [o.raw[32 * i:32 * i + 32] for i in range(n)]


def search(sp, lo, hi, key=None, mask_bits=None, cascade=None, detail=False):
    """Analyzer brute force on the CPU: (seed, [keys]) of the smallest matching seed in [lo, hi], or None.
    detail=True appends (variant, chain_index) -- meaningful for sha256_chain."""
    if cascade is not None:
        mode, n = 2, len(cascade)
        bits = bytes(b for b, _ in cascade)
        targets = (C.c_uint64 * n)(*[t for _, t in cascade])
        key = bytes(32)
    else:
        mode, n = (1 if mask_bits else 0), 1
        bits = bytes([mask_bits or 0])
        targets = (C.c_uint64 * 1)(0)
    seed, keys = C.c_uint64(), _out(32 * 8)
    vi = (C.c_uint32 * 2)()
    r = lib().vko_search(C.byref(sp), lo, hi, mode, key, n, bits, targets, C.byref(seed), keys, vi)
    if r != 1:
        return None
    ks = [keys.raw[32 * i:32 * i + 32] for i in range(n)]
    return (seed.value, ks, (vi[0], vi[1])) if detail else (seed.value, ks)


class Wordlist:
    """A caller-supplied 2048-word BIP-39 list in the form the oracle takes."""

    def __init__(self, words):
        assert len(words) == 2048
        self.words = [w.encode() if isinstance(w, str) else w for w in words]
        self.arr = (C.c_char_p * 2048)(*self.words)


def multibit_buggy_mnemonic(wl, seed64):
    o = _out(600)
    n = lib().vko_multibit_buggy_mnemonic(wl.arr, seed64, o, 600)
    return o.raw[:n].decode()


def multibit_keys(wl, mnemonic, count=20, passphrase=b""):
    """MultibitTransform::apply_batch for one string input: (seed1, seed2, [(label, key)]) or None if skipped.
    passphrase: the BIP-39 passphrase of the multibit-hd analyzer (the transform always uses the empty one)."""
    m = mnemonic.encode() if isinstance(mnemonic, str) else mnemonic
    pw = passphrase.encode() if isinstance(passphrase, str) else passphrase
    keys, valid, s1, s2 = _out(32 * count), _out(count), _out(64), _out(64)
    if not lib().vko_multibit_keys_pass(wl.arr, m, len(m), pw, len(pw), count, keys, valid, s1, s2):
        return None
    out = []
    for i in range(count):
        if valid.raw[i]:
            lab = _out(700)
            lib().vko_multibit_label(m, i, lab, 700)
            out.append((lab.value.decode(), keys.raw[32 * i:32 * i + 32]))
    return s1.raw, s2.raw, out


def base64_text(data):
    o = _out(4 * ((len(data) + 2) // 3) + 1)
    n = lib().vko_base64(data, len(data), o)
    return o.raw[:n]


def bitimage_entropy(data):
    """SHA-256 of the base64 text of the file bytes (BitimageDeriver::from_file_bytes, first half)."""
    o = _out(32)
    assert lib().vko_bitimage_entropy(data, len(data), o)
    return o.raw


def bip39_mnemonic(wl, entropy):
    o = _out(1024)
    n = lib().vko_bip39_mnemonic(wl.arr, entropy, len(entropy), o, 1024)
    return o.raw[:n].decode()


def bip32_parse_path(path):
    """Component list (index | 0x80000000 when hardened) or a negative error code (-1 prefix, -2 index, -3 limit)."""
    comps = (C.c_uint32 * 256)()
    d = lib().vko_bip32_parse_path(path.encode() if isinstance(path, str) else path, comps, 256)
    return d if d < 0 else [comps[i] for i in range(d)]


def bip32_increment_path(path):
    o = _out(600)
    n = lib().vko_bip32_increment_path(path.encode(), o, 600)
    return o.raw[:n].decode()


def bip32_derive(seed64, comps):
    arr = (C.c_uint32 * max(1, len(comps)))(*comps)
    o = _out(32)
    return o.raw if lib().vko_bip32_derive(seed64, arr, len(comps), o) else None


def bitimage_keys(wl, entropy, passphrase=b"", path="m/84'/0'/0'/0/0", count=1):
    """BitimageTransform::derive_keys_for_passphrase for one (file entropy, passphrase): [(path text, key)] in order."""
    pw = passphrase.encode() if isinstance(passphrase, str) else passphrase
    keys, valid = _out(32 * count), _out(count)
    lib().vko_bitimage_keys(wl.arr, entropy, pw, len(pw), path.encode(), count, keys, valid)
    out, cur = [], path
    for i in range(count):
        if not valid.raw[i]:
            continue
        out.append((cur, keys.raw[32 * i:32 * i + 32]))
        if i + 1 < count:
            cur = bip32_increment_path(cur)
    return out


def normalize_key(k): o = _out(32); lib().vko_normalize_key(k, o); return o.raw


def pubkey(k, slow=False):
    p33, p65 = _out(33), _out(65)
    (lib().vko_pubkey_slow if slow else lib().vko_pubkey)(k, p33, p65)
    return p33.raw, p65.raw


def pubkey_batch(keys):
    """vko_pubkey_batch: [(pub33, pub65)] for normalised keys (16-bit windows + one shared inversion)."""
    n = len(keys)
    p33, p65 = _out(33 * n), _out(65 * n)
    lib().vko_pubkey_batch.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p]
    lib().vko_pubkey_batch(b"".join(keys), n, p33, p65)
    a, b = p33.raw, p65.raw
    return [(a[33 * i:33 * i + 33], b[65 * i:65 * i + 65]) for i in range(n)]


def scalar_add(a, b):
    o = _out(32)
    ok = lib().vko_scalar_add(a, b, o)
    return o.raw if ok else None


def scalar_valid(k): return bool(lib().vko_scalar_valid(k))


def derive(k):
    d = Derived()
    lib().vko_derive(k, C.byref(d))
    return d


def derive_h160(k):
    c, u = _out(20), _out(20)
    lib().vko_derive_h160(k, c, u)
    return c.raw, u.raw


def fe_mul(a, b): o = _out(32); lib().vko_fe_mul_be(a, b, o); return o.raw
def fe_inv(a): o = _out(32); lib().vko_fe_inv_be(a, o); return o.raw


def base58check_encode(b):
    o = _out(160)
    n = lib().vko_base58check_encode(b, len(b), o, 160)
    return o.value.decode() if n >= 0 else None


def base58check_decode(s):
    o = _out(128)
    n = lib().vko_base58check_decode(s.encode(), o, 128)
    return o.raw[:n] if n >= 0 else None


def segwit_encode(hrp, ver, prog):
    o = _out(100)
    n = lib().vko_segwit_encode(hrp.encode(), ver, prog, len(prog), o, 100)
    return o.value.decode() if n >= 0 else None


def segwit_decode(hrp, addr):
    ver, prog, pl = C.c_int(), _out(40), C.c_size_t()
    ok = lib().vko_segwit_decode(hrp.encode(), addr.encode(), C.byref(ver), prog, C.byref(pl))
    return (ver.value, prog.raw[:pl.value]) if ok else None


class Matcher:
    """Matcher::from_addresses / Matcher::load (src/matcher.rs:45-66)."""

    def __init__(self, lines=()):
        self._m = lib().vko_matcher_new()
        for ln in lines:
            self.add_line(ln)

    def add_line(self, ln):
        lib().vko_matcher_add_line(self._m, ln.encode() if isinstance(ln, str) else ln)

    def count(self): return lib().vko_matcher_count(self._m)
    def check(self, derived): return lib().vko_matcher_check(self._m, C.byref(derived))
    def check_h160(self, hc, hu): return lib().vko_matcher_check_h160(self._m, hc, hu)

    def __del__(self):
        try:
            lib().vko_matcher_free(self._m)
        except Exception:
            pass


def sweep(sp, lo, hi, flags=F_COMPRESSED, matcher=None, hit_cap=4096, nthreads=1):
    """RangeSource::process on the CPU.  Returns (stats dict, [hit dict])."""
    hits = (Hit * hit_cap)()
    st = Stats()
    rc = lib().vko_sweep(C.byref(sp), lo, hi, flags, matcher._m if matcher else None, hits, hit_cap, C.byref(st), nthreads)
    if rc != 0:
        raise RuntimeError("vko_sweep failed")
    n = min(st.matches, hit_cap)
    hl = [dict(seed=h.seed, sub_index=h.sub_index, addr_kind=h.addr_kind, privkey=bytes(h.privkey), h160=bytes(h.h160))
          for h in hits[:n]]
    return dict(inputs=st.inputs, keys=st.keys, matches=st.matches, digest=tuple(st.digest)), hl
