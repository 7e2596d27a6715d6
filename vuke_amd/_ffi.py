"""ctypes binding of libvuke_hip.so (the C ABI declared in include/vuke_hip.h).

There is no fallback of any kind here: if the shared library is missing or no gfx950
device is usable, loading / engine creation raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvuke_hip.so")
LIB_PATH = os.environ.get("VUKE_HIP_LIB") or LIB_PATH  # A/B runs of two builds of the same library (tools/), nothing else
# the same sources built with -DVK_TEST_HOOKS (include/vuke_hip_test_hooks.h): tests / tools only.  Resolved on its own: an explicit
# VUKE_HIP_HOOKS_LIB, else the hooks build next to the library in use (tools/build_variant.sh links one beside every variant), else
# the in-tree hooks build -- a variant without a hooks twin must not turn Engine(hooks=True) into an OSError
_IN_TREE_HOOKS = os.path.join(_HERE, "lib", "libvuke_hip_hooks.so")
HOOKS_LIB_PATH = os.environ.get("VUKE_HIP_HOOKS_LIB") or (LIB_PATH[:-3] + "_hooks.so" if os.path.exists(LIB_PATH[:-3] + "_hooks.so") else _IN_TREE_HOOKS)

VK_OK = 0
VK_E_NO_DEVICE, VK_E_HIP, VK_E_INVALID_ARG, VK_E_UNSUPPORTED, VK_E_NOMEM = -1, -2, -3, -4, -5
T_SHA256, T_MILKSAD, T_MT64, T_LCG, T_XORSHIFT, T_SHA256_CHAIN, T_ELECTRUM, T_MULTIBIT, T_DOUBLE_SHA256, T_MD5, T_DIRECT, T_ARMORY = range(12)
VARIANT_ALL = 0xFFFFFFFF
F_COMPRESSED, F_UNCOMPRESSED = 1, 2
ADDR_P2PKH_C, ADDR_P2PKH_U, ADDR_P2WPKH = 0, 1, 2
TARGET_P2PKH, TARGET_P2WPKH = 0, 1
LAYOUT_UNIFORM, LAYOUT_SIGNED9 = 0, 1
LAYOUT_SIGNED = LAYOUT_SIGNED9    # signed-digit windows of any width (window_bits 8..29; 0 = the nine windows the old name stands for)


class TransformSpec(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("variant", C.c_uint32), ("endian", C.c_uint32), ("chain_depth", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("sub_index", C.c_uint32), ("addr_kind", C.c_uint32),
                ("privkey", C.c_uint8 * 32), ("h160", C.c_uint8 * 20), ("_pad", C.c_uint32)]


class SweepStats(C.Structure):
    _fields_ = [("inputs", C.c_uint64), ("keys", C.c_uint64), ("matches", C.c_uint64), ("hits_dropped", C.c_uint64),
                ("digest", C.c_uint64 * 4), ("kernel_seconds", C.c_double), ("launches", C.c_uint32), ("inputs_skipped", C.c_uint32),
                ("shader_clock_ghz", C.c_double)]


class EngineOpts(C.Structure):
    _fields_ = [("window_bits", C.c_uint32), ("batch", C.c_uint32), ("blocks_per_cu", C.c_uint32),
                ("layout", C.c_uint32), ("stream", C.c_void_p), ("algo", C.c_uint32), ("_pad", C.c_uint32)]


class EngineInfo(C.Structure):
    _fields_ = [("device", C.c_int32), ("cu_count", C.c_uint32), ("window_bits", C.c_uint32), ("windows", C.c_uint32),
                ("batch", C.c_uint32), ("blocks_per_cu", C.c_uint32), ("layout", C.c_uint32), ("_pad", C.c_uint32),
                ("table_bytes", C.c_uint64),
                ("table_build_seconds", C.c_double), ("runtime_init_seconds", C.c_double), ("alloc_seconds", C.c_double),
                ("table_wall_seconds", C.c_double)]


class SearchTarget(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("n_cascade", C.c_uint32), ("key", C.c_uint8 * 32), ("mask_bits", C.c_uint32), ("_pad", C.c_uint32),
                ("bits", C.c_uint8 * 8), ("targets", C.c_uint64 * 8)]


class SearchResult(C.Structure):
    _fields_ = [("found", C.c_uint32), ("n_keys", C.c_uint32), ("seed", C.c_uint64), ("seeds_checked", C.c_uint64),
                ("kernel_seconds", C.c_double), ("keys", (C.c_uint8 * 32) * 8), ("variant", C.c_uint32), ("index", C.c_uint32)]


RC_COLUMNS = 15
RC_NAMES = ("source", "private_key_raw", "private_key_hex", "private_key_decimal", "private_key_binary", "private_key_bit_length",
            "private_key_hamming_weight", "private_key_leading_zeros", "pubkey_compressed", "pubkey_uncompressed", "address_p2pkh_compressed",
            "address_p2pkh_uncompressed", "address_p2wpkh", "wif_compressed", "wif_uncompressed")   # VK_RC_* in order


class RecordColumns(C.Structure):
    _fields_ = [("data", C.c_void_p * RC_COLUMNS), ("cap", C.c_size_t * RC_COLUMNS), ("len", C.c_size_t * RC_COLUMNS), ("rows", C.c_uint64)]


SEARCH_EXACT, SEARCH_MASKED, SEARCH_CASCADE = 0, 1, 2

EXPORTS = [
    "vk_abi_version", "vk_device_count", "vk_engine_create", "vk_engine_create_ex", "vk_engine_destroy", "vk_engine_info",
    "vk_last_error", "vk_last_warning", "vk_targets_set", "vk_sweep", "vk_generate", "vk_generate_lines", "vk_generate_verbose", "vk_generate_lines_strings", "vk_generate_columns", "vk_derive_columns", "vk_record_column_stride", "vk_derive_batch", "vk_derive_batch_pub", "vk_transform_batch",
    "vk_transform_strings", "vk_bitimage_keys", "vk_scan_keys", "vk_scan_strings", "vk_rescan_strings", "vk_bip39_wordlist_set", "vk_multibit_passphrase_set", "vk_search",
    "vk_slots_per_input", "vk_slots_per_string",
]
HOOK_EXPORTS = ["vk_debug_table_entry", "vk_debug_field", "vk_debug_scratch", "vk_debug_chain_stop", "vk_debug_ring_knobs", "vk_debug_scan", "vk_debug_last_launch", "vk_debug_fail_alloc"]

_lib = None
_hooks_lib = None


def load(hooks=False):
    """Load libvuke_hip.so (hooks=True: libvuke_hip_hooks.so, the test build that also exports the vk_debug_* hooks); raises
    OSError if it has not been built (no fallback)."""
    global _lib, _hooks_lib
    if not hooks and _lib is not None:
        return _lib
    if hooks and _hooks_lib is not None:
        return _hooks_lib
    path = HOOKS_LIB_PATH if hooks else LIB_PATH
    if not os.path.exists(path):
        raise OSError(f"{path} is missing: build it with vuke_amd/csrc/build.sh (or __graft_entry__.build())")
    L = C.CDLL(path)
    vp, cp, sz, u32, u64 = C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint64
    L.vk_abi_version.restype = C.c_int
    L.vk_device_count.restype = C.c_int
    L.vk_engine_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.vk_engine_create_ex.argtypes = [C.c_int, C.POINTER(EngineOpts), C.POINTER(vp)]
    L.vk_engine_destroy.argtypes = [vp]
    L.vk_engine_destroy.restype = None
    L.vk_engine_info.argtypes = [vp, C.POINTER(EngineInfo)]
    L.vk_last_error.argtypes = [vp]
    L.vk_last_error.restype = cp
    L.vk_last_warning.argtypes = [vp, vp, sz]
    L.vk_last_warning.restype = sz
    L.vk_targets_set.argtypes = [vp, cp, cp, sz]
    L.vk_sweep.argtypes = [vp, C.POINTER(TransformSpec), u64, u64, u32, C.POINTER(Hit), sz, C.POINTER(SweepStats)]
    L.vk_generate.argtypes = [vp, C.POINTER(TransformSpec), u64, u64, u32, vp, vp, vp, vp, C.POINTER(SweepStats)]
    L.vk_generate_lines.argtypes = [vp, C.POINTER(TransformSpec), u64, u64, cp, C.POINTER(C.c_char_p), vp, sz, C.POINTER(sz), C.POINTER(SweepStats)]
    L.vk_generate_verbose.argtypes = L.vk_generate_lines.argtypes
    L.vk_generate_lines_strings.argtypes = [vp, C.POINTER(TransformSpec), cp, C.POINTER(u32), sz, cp, C.POINTER(u32), cp, C.POINTER(C.c_char_p), vp, sz, C.POINTER(sz),
                                            C.POINTER(SweepStats)]
    L.vk_derive_columns.argtypes = [vp, cp, sz, C.POINTER(RecordColumns)]
    L.vk_record_column_stride.restype = sz
    L.vk_record_column_stride.argtypes = [C.c_int]
    L.vk_generate_columns.argtypes = [vp, C.POINTER(TransformSpec), u64, u64, C.POINTER(C.c_char_p), C.POINTER(RecordColumns), C.POINTER(SweepStats)]
    L.vk_derive_batch.argtypes = [vp, cp, sz, vp, vp, vp]
    L.vk_derive_batch_pub.argtypes = [vp, cp, sz, vp, vp, vp, vp]
    L.vk_transform_batch.argtypes = [vp, C.POINTER(TransformSpec), C.POINTER(u64), sz, vp, vp]
    L.vk_transform_strings.argtypes = [vp, C.POINTER(TransformSpec), cp, C.POINTER(u32), sz, vp, vp]
    L.vk_rescan_strings.argtypes = [vp, u32, C.POINTER(Hit), sz, C.POINTER(SweepStats)]
    L.vk_bip39_wordlist_set.argtypes = [vp, cp]
    L.vk_bitimage_keys.argtypes = [vp, cp, sz, cp, C.POINTER(u32), sz, C.POINTER(u32), u32, u32, cp, cp]
    L.vk_search.argtypes = [vp, C.POINTER(TransformSpec), u64, u64, C.POINTER(SearchTarget), C.POINTER(SearchResult)]
    L.vk_scan_keys.argtypes = [vp, cp, sz, u32, C.POINTER(Hit), sz, C.POINTER(SweepStats)]
    L.vk_scan_strings.argtypes = [vp, C.POINTER(TransformSpec), cp, C.POINTER(u32), sz, u32, C.POINTER(Hit), sz, C.POINTER(SweepStats)]
    L.vk_slots_per_input.argtypes = [C.POINTER(TransformSpec)]
    L.vk_slots_per_input.restype = u32
    L.vk_slots_per_string.argtypes = [C.POINTER(TransformSpec)]
    L.vk_slots_per_string.restype = u32
    if hooks:
        L.vk_debug_table_entry.argtypes = [vp, u32, u32, vp]
        L.vk_debug_field.argtypes = [vp, C.c_int, cp, cp, sz, vp]
        L.vk_debug_scratch.argtypes = [vp, C.c_int, sz, vp, sz]
        L.vk_debug_chain_stop.argtypes = [vp, C.c_int]
        L.vk_debug_ring_knobs.argtypes = [vp, u32, u32, u32]
        L.vk_debug_scan.argtypes = [vp, vp, sz, vp]
        L.vk_debug_last_launch.argtypes = [vp, vp]
        L.vk_debug_fail_alloc.argtypes = [C.c_longlong]
        L.vk_debug_fail_alloc.restype = C.c_longlong
        _hooks_lib = L
    else:
        _lib = L
    return L


class VkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vuke_hip error {code}: {msg}")
        self.code = code


class Engine:
    """One engine per GPU (vk_engine).  Raises VkError when no gfx950 device is usable."""

    def __init__(self, device=0, window_bits=0, batch=0, blocks_per_cu=0, stream=None, layout=0, algo=0, hooks=False):
        L = load(hooks)     # hooks=True: the test build (vk_debug_* available); the product never asks for it
        self._L = L
        self._e = C.c_void_p()
        opts = EngineOpts(window_bits, batch, blocks_per_cu, layout, stream, algo, 0)
        rc = L.vk_engine_create_ex(device, C.byref(opts), C.byref(self._e))
        if rc != VK_OK:
            raise VkError(rc, (L.vk_last_error(None) or b"").decode())

    def close(self):
        if getattr(self, "_e", None) and self._e.value:
            self._L.vk_engine_destroy(self._e)
            self._e = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = EngineInfo()
        self._chk(self._L.vk_engine_info(self._e, C.byref(i)))
        return {k: getattr(i, k) for k, _ in EngineInfo._fields_}

    def _chk(self, rc):
        if rc != VK_OK:
            raise VkError(rc, (self._L.vk_last_error(self._e) or b"").decode())

    def targets_set(self, h160s, kinds):
        n = len(h160s)
        self._chk(self._L.vk_targets_set(self._e, b"".join(h160s), bytes(kinds), n))

    def sweep(self, spec, lo, hi, flags=F_COMPRESSED, hit_cap=4096):
        hits = (Hit * max(hit_cap, 1))()
        st = SweepStats()
        self._chk(self._L.vk_sweep(self._e, C.byref(spec), lo, hi, flags, hits, hit_cap, C.byref(st)))
        n = min(st.matches, hit_cap)
        hl = [dict(seed=h.seed, sub_index=h.sub_index, addr_kind=h.addr_kind, privkey=bytes(h.privkey), h160=bytes(h.h160))
              for h in hits[:n]]
        hl.sort(key=lambda h: (h["seed"], h["sub_index"]))
        return dict(inputs=st.inputs, keys=st.keys, matches=st.matches, hits_dropped=st.hits_dropped,
                    digest=tuple(st.digest), kernel_seconds=st.kernel_seconds, launches=st.launches, shader_clock_ghz=st.shader_clock_ghz), hl

    def generate(self, spec, lo, hi, flags=F_COMPRESSED | F_UNCOMPRESSED):
        slots = self._L.vk_slots_per_input(C.byref(spec))
        n = (hi - lo + 1) * slots
        priv, hc, hu, valid = (C.create_string_buffer(n * 32), C.create_string_buffer(n * 20),
                               C.create_string_buffer(n * 20), C.create_string_buffer(n))
        st = SweepStats()
        self._chk(self._L.vk_generate(self._e, C.byref(spec), lo, hi, flags, priv, hc, hu, valid, C.byref(st)))
        return dict(slots=slots, n=n, privkeys=priv.raw, h160_c=hc.raw, h160_u=hu.raw, valid=valid.raw,
                    keys=st.keys, inputs=st.inputs, digest=tuple(st.digest), kernel_seconds=st.kernel_seconds)

    def generate_lines(self, spec, lo, hi, name, suffixes=None, verbose=False):
        """Compact CSV text (vk_generate_lines) or verbose blocks (vk_generate_verbose) of generate mode for [lo, hi]: (bytes, stats)."""
        slots = self._L.vk_slots_per_input(C.byref(spec))
        n = (hi - lo + 1) * slots
        buf = C.create_string_buffer(max(1, n * (512 if verbose else 192)))
        ln = C.c_size_t(0)
        st = SweepStats()
        arr = None
        if suffixes is not None:
            arr = (C.c_char_p * slots)(*[s if isinstance(s, bytes) else s.encode() for s in suffixes])
        fn = self._L.vk_generate_verbose if verbose else self._L.vk_generate_lines
        self._chk(fn(self._e, C.byref(spec), lo, hi, name if isinstance(name, bytes) else name.encode(), arr, buf, len(buf), C.byref(ln), C.byref(st)))
        return buf.raw[:ln.value], dict(keys=st.keys, inputs=st.inputs, kernel_seconds=st.kernel_seconds)

    def _column_buffers(self, items):
        rc = RecordColumns()
        bufs = []
        for c in range(RC_COLUMNS):
            b = C.create_string_buffer(max(1, items * self._L.vk_record_column_stride(c)))
            bufs.append(b)
            rc.data[c] = C.cast(b, C.c_void_p)
            rc.cap[c] = len(b)
        return rc, bufs

    def derive_columns(self, keys):
        """KeyDeriver::derive as columns (vk_derive_columns): ({column name: PLAIN page bytes}, rows)."""
        rc, bufs = self._column_buffers(len(keys))
        self._chk(self._L.vk_derive_columns(self._e, b"".join(keys), len(keys), C.byref(rc)))
        return {RC_NAMES[c]: bufs[c].raw[:rc.len[c]] for c in range(RC_COLUMNS)}, rc.rows

    def generate_columns(self, spec, lo, hi, suffixes=None):
        """The value columns of the full record (vk_generate_columns): ({column name: PLAIN page bytes}, rows, stats)."""
        slots = self._L.vk_slots_per_input(C.byref(spec))
        rc, bufs = self._column_buffers((hi - lo + 1) * slots)
        arr = None
        if suffixes is not None:
            arr = (C.c_char_p * slots)(*[s if isinstance(s, bytes) else s.encode() for s in suffixes])
        st = SweepStats()
        self._chk(self._L.vk_generate_columns(self._e, C.byref(spec), lo, hi, arr, C.byref(rc), C.byref(st)))
        return ({RC_NAMES[c]: bufs[c].raw[:rc.len[c]] for c in range(RC_COLUMNS)}, rc.rows,
                dict(keys=st.keys, inputs=st.inputs, kernel_seconds=st.kernel_seconds))

    def generate_lines_strings(self, spec, strings, name, suffixes=None, labels=None):
        """Compact CSV text of generate mode for string inputs (vk_generate_lines_strings): (bytes, stats)."""
        blob, offs, n = pack_strings(strings)
        lblob, loffs = None, None
        if labels is not None:
            lblob, loffs, _ = pack_strings(labels)
        slots = self._L.vk_slots_per_string(C.byref(spec))
        buf = C.create_string_buffer(max(1, len(blob if lblob is None else lblob) * slots + n * slots * 192))
        ln = C.c_size_t(0)
        st = SweepStats()
        arr = None
        if suffixes is not None:
            arr = (C.c_char_p * slots)(*[s if isinstance(s, bytes) else s.encode() for s in suffixes])
        self._chk(self._L.vk_generate_lines_strings(self._e, C.byref(spec), blob, offs, n, lblob, loffs, name if isinstance(name, bytes) else name.encode(), arr, buf,
                                                     len(buf), C.byref(ln), C.byref(st)))
        return buf.raw[:ln.value], dict(keys=st.keys, inputs=st.inputs, kernel_seconds=st.kernel_seconds)

    def derive_batch(self, keys):
        n = len(keys)
        norm, hc, hu = C.create_string_buffer(n * 32), C.create_string_buffer(n * 20), C.create_string_buffer(n * 20)
        self._chk(self._L.vk_derive_batch(self._e, b"".join(keys), n, norm, hc, hu))
        rn, rc, ru = norm.raw, hc.raw, hu.raw  # .raw copies the whole buffer: take it once
        return ([rn[32 * i:32 * i + 32] for i in range(n)], [rc[20 * i:20 * i + 20] for i in range(n)],
                [ru[20 * i:20 * i + 20] for i in range(n)])

    def derive_pubkeys(self, keys):
        """KeyDeriver::derive with the public keys: per key (normalised, pub33, pub65, h160_c, h160_u)."""
        n = len(keys)
        norm, hc, hu, xy = (C.create_string_buffer(n * 32), C.create_string_buffer(n * 20), C.create_string_buffer(n * 20),
                            C.create_string_buffer(n * 64))
        self._chk(self._L.vk_derive_batch_pub(self._e, b"".join(keys), n, norm, hc, hu, xy))
        rn, rc, ru, rx = norm.raw, hc.raw, hu.raw, xy.raw
        out = []
        for i in range(n):
            x, y = rx[64 * i:64 * i + 32], rx[64 * i + 32:64 * i + 64]
            out.append((rn[32 * i:32 * i + 32], bytes([2 | (y[31] & 1)]) + x, b"\x04" + x + y, rc[20 * i:20 * i + 20], ru[20 * i:20 * i + 20]))
        return out

    def transform_batch(self, spec, seeds):
        n = len(seeds)
        slots = self._L.vk_slots_per_input(C.byref(spec))
        arr = (C.c_uint64 * n)(*seeds)
        keys, valid = C.create_string_buffer(n * slots * 32), C.create_string_buffer(n * slots)
        self._chk(self._L.vk_transform_batch(self._e, C.byref(spec), arr, n, keys, valid))
        rk, rv = keys.raw, valid.raw
        return [[(rk[32 * (i * slots + s):32 * (i * slots + s) + 32] if rv[i * slots + s] else None) for s in range(slots)]
                for i in range(n)]

    def bip39_wordlist_set(self, words):
        """words: 2048 byte strings of <= 8 bytes each."""
        assert len(words) == 2048
        blob = b"".join((w if isinstance(w, bytes) else w.encode()).ljust(8, b"\0") for w in words)
        self._chk(self._L.vk_bip39_wordlist_set(self._e, blob))

    def multibit_passphrase_set(self, passphrase=b""):
        pw = passphrase.encode() if isinstance(passphrase, str) else passphrase
        self._L.vk_multibit_passphrase_set.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        self._chk(self._L.vk_multibit_passphrase_set(self._e, pw, len(pw)))

    def bitimage_keys(self, entropies, passphrases=(b"",), path=(84 | 1 << 31, 1 << 31, 1 << 31, 0, 0), count=1):
        """BitimageTransform::apply_batch after the files have been reduced to their entropies (SHA-256 of the base64
        text): result[file][passphrase] = list of `count` keys (None where the reference emits nothing).
        path = components, index | 0x80000000 when hardened."""
        nf, npw = len(entropies), len(passphrases)
        offs = (C.c_uint32 * (npw + 1))()
        for i, s in enumerate(passphrases):
            offs[i + 1] = offs[i] + len(s)
        comps = (C.c_uint32 * max(1, len(path)))(*path)
        total = nf * npw * count
        keys, valid = C.create_string_buffer(max(1, total * 32)), C.create_string_buffer(max(1, total))
        self._chk(self._L.vk_bitimage_keys(self._e, b"".join(entropies), nf, b"".join(passphrases), offs, npw, comps, len(path), count, keys, valid))
        rk, rv = keys.raw, valid.raw
        out = []
        for f in range(nf):
            row = []
            for j in range(npw):
                base = (f * npw + j) * count
                row.append([rk[32 * (base + c):32 * (base + c) + 32] if rv[base + c] else None for c in range(count)])
            out.append(row)
        return out

    def transform_strings(self, spec, strings, raw=False):
        """Transform::apply_batch for string inputs: per input a list of keys (None where nothing is emitted).
        raw=True returns (keys bytes, valid bytes, seconds inside the library call) without per-key unpacking."""
        n = len(strings)
        slots = self._L.vk_slots_per_string(C.byref(spec))
        offs = (C.c_uint32 * (n + 1))()
        for i, s in enumerate(strings):
            offs[i + 1] = offs[i] + len(s)
        keys, valid = C.create_string_buffer(max(1, n * slots * 32)), C.create_string_buffer(max(1, n * slots))
        blob = b"".join(strings)
        import time
        t0 = time.perf_counter()
        self._chk(self._L.vk_transform_strings(self._e, C.byref(spec), blob, offs, n, keys, valid))
        dt = time.perf_counter() - t0
        rk, rv = keys.raw, valid.raw
        if raw:
            return rk, rv, dt
        return [[(rk[32 * (i * slots + s):32 * (i * slots + s) + 32] if rv[i * slots + s] else None) for s in range(slots)]
                for i in range(n)]

    def debug_chain_stop(self, stages):
        self._chk(self._L.vk_debug_chain_stop(self._e, stages))

    def debug_scan(self, values):
        """Exclusive prefix sum mod 2^32 of a numpy uint32 array through vk_scan_kernel; hooks build only."""
        import numpy as np
        a = np.ascontiguousarray(values, dtype=np.uint32)
        out = np.empty_like(a)
        self._chk(self._L.vk_debug_scan(self._e, a.ctypes.data, a.size, out.ctypes.data))
        return out

    def debug_last_launch(self):
        """{form, grid, block, R} of the engine's last sweep launch (form: 2 two waves, 3 three waves compressed-only, 4 the
        768-lane workgroup per CU of launches that hash uncompressed keys); hooks build only."""
        out = (C.c_uint32 * 4)()
        self._chk(self._L.vk_debug_last_launch(self._e, out))
        return {"form": out[0], "grid": out[1], "block": out[2], "R": out[3]}

    def debug_ring_knobs(self, chunk_iters=0, waves_per_simd=0, workers=0):
        """Launch policy of the chain ring for the following chain calls (0 = the library's choice); hooks build only."""
        self._chk(self._L.vk_debug_ring_knobs(self._e, chunk_iters, waves_per_simd, workers))

    def debug_scratch(self, slot, offset, n):
        out = C.create_string_buffer(n)
        self._chk(self._L.vk_debug_scratch(self._e, slot, offset, out, n))
        return out.raw

    def scan_keys(self, keys, flags=F_COMPRESSED | F_UNCOMPRESSED, hit_cap=4096):
        hits = (Hit * max(hit_cap, 1))()
        st = SweepStats()
        self._chk(self._L.vk_scan_keys(self._e, b"".join(keys), len(keys), flags, hits, hit_cap, C.byref(st)))
        n = min(st.matches, hit_cap)
        hl = sorted((dict(index=h.seed, addr_kind=h.addr_kind, privkey=bytes(h.privkey), h160=bytes(h.h160)) for h in hits[:n]),
                    key=lambda h: h["index"])
        return dict(inputs=st.inputs, keys=st.keys, matches=st.matches, digest=tuple(st.digest)), hl

    def scan_strings(self, spec, strings, flags=F_COMPRESSED | F_UNCOMPRESSED, hit_cap=4096, packed=None):
        """Wordlist / stdin source in scan mode: strings -> transform -> derive -> probe without leaving the device.
        packed = (blob, offsets ctypes array, n) skips the per-call packing (bench)."""
        if packed is None:
            packed = pack_strings(strings)
        blob, offs, n = packed
        hits = (Hit * max(hit_cap, 1))()
        st = SweepStats()
        self._chk(self._L.vk_scan_strings(self._e, C.byref(spec), blob, offs, n, flags, hits, hit_cap, C.byref(st)))
        m = min(st.matches, hit_cap)
        hl = sorted((dict(index=h.seed, sub_index=h.sub_index, addr_kind=h.addr_kind, privkey=bytes(h.privkey), h160=bytes(h.h160))
                     for h in hits[:m]), key=lambda h: (h["index"], h["sub_index"]))
        warn = self.last_warning()
        return dict(inputs=st.inputs, inputs_skipped=st.inputs_skipped, keys=st.keys, matches=st.matches, hits_dropped=st.hits_dropped, digest=tuple(st.digest),
                    kernel_seconds=st.kernel_seconds, launches=st.launches, shader_clock_ghz=st.shader_clock_ghz, warning=warn), hl

    def rescan_strings(self, flags=F_COMPRESSED | F_UNCOMPRESSED, hit_cap=4096):
        """The probe sweep of the last scan_strings call again (its keys are still on the device), with another hit buffer."""
        hits = (Hit * max(hit_cap, 1))()
        st = SweepStats()
        self._chk(self._L.vk_rescan_strings(self._e, flags, hits, hit_cap, C.byref(st)))
        m = min(st.matches, hit_cap)
        hl = sorted((dict(index=h.seed, sub_index=h.sub_index, addr_kind=h.addr_kind, privkey=bytes(h.privkey), h160=bytes(h.h160))
                     for h in hits[:m]), key=lambda h: (h["index"], h["sub_index"]))
        return dict(inputs=st.inputs, keys=st.keys, matches=st.matches, hits_dropped=st.hits_dropped, digest=tuple(st.digest)), hl

    def last_warning(self):
        buf = C.create_string_buffer(512)
        self._L.vk_last_warning(self._e, buf, len(buf))
        return buf.value.decode()

    def search(self, spec, lo, hi, key=None, mask_bits=None, cascade=None):
        """Analyzer brute force: smallest matching seed in [lo, hi] (exact key, masked key, or cascade [(bits, target)])."""
        tg = SearchTarget()
        if cascade is not None:
            tg.mode, tg.n_cascade = SEARCH_CASCADE, len(cascade)
            for j, (b, t) in enumerate(cascade):
                tg.bits[j], tg.targets[j] = b, t
        else:
            tg.mode = SEARCH_MASKED if mask_bits else SEARCH_EXACT
            tg.mask_bits = mask_bits or 0
            tg.key[:] = key
        r = SearchResult()
        self._chk(self._L.vk_search(self._e, C.byref(spec), lo, hi, C.byref(tg), C.byref(r)))
        return dict(found=bool(r.found), seed=r.seed, seeds_checked=r.seeds_checked, kernel_seconds=r.kernel_seconds,
                    keys=[bytes(r.keys[j]) for j in range(r.n_keys)], variant=r.variant, index=r.index)

    def table_entry(self, window, digit):
        xy = C.create_string_buffer(64)
        self._chk(self._L.vk_debug_table_entry(self._e, window, digit, xy))
        return xy.raw[:32], xy.raw[32:]

    def field_op(self, op, a_list, b_list=None):
        n = len(a_list)
        out = C.create_string_buffer(n * 32)
        self._chk(self._L.vk_debug_field(self._e, op, b"".join(a_list), b"".join(b_list) if b_list else None, n, out))
        ro = out.raw
        return [ro[32 * i:32 * i + 32] for i in range(n)]


def pack_strings(strings):
    """(blob, offsets, n) as vk_transform_strings / vk_scan_strings take them."""
    n = len(strings)
    offs = (C.c_uint32 * (n + 1))()
    for i, s in enumerate(strings):
        offs[i + 1] = offs[i] + len(s)
    return b"".join(strings), offs, n


def spec(kind, variant=VARIANT_ALL, endian=0, chain_depth=10):
    return TransformSpec(kind, variant, endian, chain_depth)
