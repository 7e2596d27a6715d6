//! 1:1 with `include/vuke_hip.h` (ABI version 3).  Every `#[repr(C)]` struct carries compile-time layout assertions; the
//! same numbers are checked against the C header by `tests/test_abi_layout.py` in the engine's repository.
use core::mem::{offset_of, size_of};
use std::os::raw::{c_char, c_int, c_void};

pub const VK_ABI_VERSION: c_int = 5;
pub const VK_OK: c_int = 0;
pub const VK_E_NO_DEVICE: c_int = -1;
pub const VK_E_HIP: c_int = -2;
pub const VK_E_INVALID_ARG: c_int = -3;
pub const VK_E_UNSUPPORTED: c_int = -4;
pub const VK_E_NOMEM: c_int = -5;

pub const VK_T_SHA256: u32 = 0;
pub const VK_T_MILKSAD: u32 = 1;
pub const VK_T_MT64: u32 = 2;
pub const VK_T_LCG: u32 = 3;
pub const VK_T_XORSHIFT: u32 = 4;
pub const VK_T_SHA256_CHAIN: u32 = 5;
pub const VK_T_ELECTRUM: u32 = 6;
pub const VK_T_MULTIBIT: u32 = 7;
pub const VK_T_DOUBLE_SHA256: u32 = 8;
pub const VK_T_MD5: u32 = 9;
pub const VK_T_DIRECT: u32 = 10;
pub const VK_T_ARMORY: u32 = 11;
pub const VK_T_BITIMAGE: u32 = 12;
pub const VK_VARIANT_ALL: u32 = 0xFFFF_FFFF;
pub const VK_F_COMPRESSED: u32 = 1;
pub const VK_F_UNCOMPRESSED: u32 = 2;
pub const VK_TARGET_P2PKH: u8 = 0;
pub const VK_TARGET_P2WPKH: u8 = 1;
pub const VK_ADDR_P2PKH_C: u32 = 0;
pub const VK_ADDR_P2PKH_U: u32 = 1;
pub const VK_ADDR_P2WPKH: u32 = 2;
// vk_engine_opts.layout / .algo, vk_search_target.mode (found missing by tests/test_abi_layout.py in round 6: the crate spelled them as
// bare numbers in its doc comments)
pub const VK_LAYOUT_UNIFORM: u32 = 0;
pub const VK_LAYOUT_SIGNED: u32 = 1;
pub const VK_LAYOUT_SIGNED9: u32 = 1;
pub const VK_ALGO_CHAIN: u32 = 0;
pub const VK_SEARCH_EXACT: u32 = 0;
pub const VK_SEARCH_MASKED: u32 = 1;
pub const VK_SEARCH_CASCADE: u32 = 2;
pub const VK_RC_SOURCE: usize = 0;
pub const VK_RC_KEY_RAW: usize = 1;
pub const VK_RC_KEY_HEX: usize = 2;
pub const VK_RC_KEY_DECIMAL: usize = 3;
pub const VK_RC_KEY_BINARY: usize = 4;
pub const VK_RC_BIT_LENGTH: usize = 5;
pub const VK_RC_HAMMING_WEIGHT: usize = 6;
pub const VK_RC_LEADING_ZEROS: usize = 7;
pub const VK_RC_PUBKEY_C: usize = 8;
pub const VK_RC_PUBKEY_U: usize = 9;
pub const VK_RC_P2PKH_C: usize = 10;
pub const VK_RC_P2PKH_U: usize = 11;
pub const VK_RC_P2WPKH: usize = 12;
pub const VK_RC_WIF_C: usize = 13;
pub const VK_RC_WIF_U: usize = 14;
pub const VK_RC_COLUMNS: usize = 15;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct VkTransformSpec {
    pub kind: u32,
    pub variant: u32,
    pub endian: u32,
    pub chain_depth: u32,
}
const _: () = assert!(size_of::<VkTransformSpec>() == 16);
const _: () = assert!(offset_of!(VkTransformSpec, kind) == 0);
const _: () = assert!(offset_of!(VkTransformSpec, variant) == 4);
const _: () = assert!(offset_of!(VkTransformSpec, endian) == 8);
const _: () = assert!(offset_of!(VkTransformSpec, chain_depth) == 12);

#[repr(C)]
#[derive(Clone, Copy)]
pub struct VkHit {
    pub seed: u64,
    pub sub_index: u32,
    pub addr_kind: u32,
    pub privkey: [u8; 32],
    pub h160: [u8; 20],
    pub _pad: u32,
}
const _: () = assert!(size_of::<VkHit>() == 72);
const _: () = assert!(offset_of!(VkHit, seed) == 0);
const _: () = assert!(offset_of!(VkHit, sub_index) == 8);
const _: () = assert!(offset_of!(VkHit, addr_kind) == 12);
const _: () = assert!(offset_of!(VkHit, privkey) == 16);
const _: () = assert!(offset_of!(VkHit, h160) == 48);
const _: () = assert!(offset_of!(VkHit, _pad) == 68);

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct VkSweepStats {
    pub inputs: u64,
    pub keys: u64,
    pub matches: u64,
    pub hits_dropped: u64,
    pub digest: [u64; 4],
    pub kernel_seconds: f64,
    pub launches: u32,
    pub inputs_skipped: u32,
    pub shader_clock_ghz: f64,
}
const _: () = assert!(size_of::<VkSweepStats>() == 88);
const _: () = assert!(offset_of!(VkSweepStats, inputs) == 0);
const _: () = assert!(offset_of!(VkSweepStats, keys) == 8);
const _: () = assert!(offset_of!(VkSweepStats, matches) == 16);
const _: () = assert!(offset_of!(VkSweepStats, hits_dropped) == 24);
const _: () = assert!(offset_of!(VkSweepStats, digest) == 32);
const _: () = assert!(offset_of!(VkSweepStats, kernel_seconds) == 64);
const _: () = assert!(offset_of!(VkSweepStats, launches) == 72);
const _: () = assert!(offset_of!(VkSweepStats, inputs_skipped) == 76);
const _: () = assert!(offset_of!(VkSweepStats, shader_clock_ghz) == 80);

#[repr(C)]
pub struct VkEngineOpts {
    pub window_bits: u32,
    pub batch: u32,
    pub blocks_per_cu: u32,
    pub layout: u32,
    pub stream: *mut c_void,
    pub algo: u32,
    pub _pad: u32,
}
const _: () = assert!(size_of::<VkEngineOpts>() == 32);
const _: () = assert!(offset_of!(VkEngineOpts, window_bits) == 0);
const _: () = assert!(offset_of!(VkEngineOpts, batch) == 4);
const _: () = assert!(offset_of!(VkEngineOpts, blocks_per_cu) == 8);
const _: () = assert!(offset_of!(VkEngineOpts, layout) == 12);
const _: () = assert!(offset_of!(VkEngineOpts, stream) == 16);
const _: () = assert!(offset_of!(VkEngineOpts, algo) == 24);
const _: () = assert!(offset_of!(VkEngineOpts, _pad) == 28);

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct VkEngineInfo {
    pub device: i32,
    pub cu_count: u32,
    pub window_bits: u32,
    pub windows: u32,
    pub batch: u32,
    pub blocks_per_cu: u32,
    pub layout: u32,
    pub _pad: u32,
    pub table_bytes: u64,
    pub table_build_seconds: f64,
    pub runtime_init_seconds: f64,
    pub alloc_seconds: f64,
    pub table_wall_seconds: f64,
}
const _: () = assert!(size_of::<VkEngineInfo>() == 72);
const _: () = assert!(offset_of!(VkEngineInfo, device) == 0);
const _: () = assert!(offset_of!(VkEngineInfo, cu_count) == 4);
const _: () = assert!(offset_of!(VkEngineInfo, window_bits) == 8);
const _: () = assert!(offset_of!(VkEngineInfo, windows) == 12);
const _: () = assert!(offset_of!(VkEngineInfo, batch) == 16);
const _: () = assert!(offset_of!(VkEngineInfo, blocks_per_cu) == 20);
const _: () = assert!(offset_of!(VkEngineInfo, layout) == 24);
const _: () = assert!(offset_of!(VkEngineInfo, _pad) == 28);
const _: () = assert!(offset_of!(VkEngineInfo, table_bytes) == 32);
const _: () = assert!(offset_of!(VkEngineInfo, table_build_seconds) == 40);
const _: () = assert!(offset_of!(VkEngineInfo, runtime_init_seconds) == 48);
const _: () = assert!(offset_of!(VkEngineInfo, alloc_seconds) == 56);
const _: () = assert!(offset_of!(VkEngineInfo, table_wall_seconds) == 64);

/// AnalysisConfig of the reference (`src/analyze/mod.rs:61-66`) in device form.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct VkSearchTarget {
    pub mode: u32,
    pub n_cascade: u32,
    pub key: [u8; 32],
    pub mask_bits: u32,
    pub _pad: u32,
    pub bits: [u8; 8],
    pub targets: [u64; 8],
}
const _: () = assert!(size_of::<VkSearchTarget>() == 120);
const _: () = assert!(offset_of!(VkSearchTarget, mode) == 0);
const _: () = assert!(offset_of!(VkSearchTarget, n_cascade) == 4);
const _: () = assert!(offset_of!(VkSearchTarget, key) == 8);
const _: () = assert!(offset_of!(VkSearchTarget, mask_bits) == 40);
const _: () = assert!(offset_of!(VkSearchTarget, _pad) == 44);
const _: () = assert!(offset_of!(VkSearchTarget, bits) == 48);
const _: () = assert!(offset_of!(VkSearchTarget, targets) == 56);

#[repr(C)]
#[derive(Clone, Copy)]
pub struct VkSearchResult {
    pub found: u32,
    pub n_keys: u32,
    pub seed: u64,
    pub seeds_checked: u64,
    pub kernel_seconds: f64,
    pub keys: [[u8; 32]; 8],
    pub variant: u32,
    pub index: u32,
}
const _: () = assert!(size_of::<VkSearchResult>() == 296);
const _: () = assert!(offset_of!(VkSearchResult, found) == 0);
const _: () = assert!(offset_of!(VkSearchResult, n_keys) == 4);
const _: () = assert!(offset_of!(VkSearchResult, seed) == 8);
const _: () = assert!(offset_of!(VkSearchResult, seeds_checked) == 16);
const _: () = assert!(offset_of!(VkSearchResult, kernel_seconds) == 24);
const _: () = assert!(offset_of!(VkSearchResult, keys) == 32);
const _: () = assert!(offset_of!(VkSearchResult, variant) == 288);
const _: () = assert!(offset_of!(VkSearchResult, index) == 292);

/// vk_record_columns: the value columns of generate mode's full record as Parquet PLAIN page bytes (vk_generate_columns)
#[repr(C)]
pub struct VkRecordColumns {
    pub data: [*mut u8; VK_RC_COLUMNS],
    pub cap: [usize; VK_RC_COLUMNS],
    pub len: [usize; VK_RC_COLUMNS],
    pub rows: u64,
}
const _: () = assert!(size_of::<VkRecordColumns>() == 368);
const _: () = assert!(offset_of!(VkRecordColumns, data) == 0);
const _: () = assert!(offset_of!(VkRecordColumns, cap) == 120);
const _: () = assert!(offset_of!(VkRecordColumns, len) == 240);
const _: () = assert!(offset_of!(VkRecordColumns, rows) == 360);

pub enum VkEngine {}

extern "C" {
    pub fn vk_abi_version() -> c_int;
    pub fn vk_device_count() -> c_int;
    pub fn vk_engine_create(device: c_int, out: *mut *mut VkEngine) -> c_int;
    pub fn vk_engine_create_ex(device: c_int, opts: *const VkEngineOpts, out: *mut *mut VkEngine) -> c_int;
    pub fn vk_engine_destroy(e: *mut VkEngine);
    pub fn vk_engine_info(e: *const VkEngine, out: *mut VkEngineInfo) -> c_int;
    pub fn vk_last_error(e: *const VkEngine) -> *const c_char;
    pub fn vk_last_warning(e: *mut VkEngine, buf: *mut c_char, cap: usize) -> usize;
    pub fn vk_targets_set(e: *mut VkEngine, h160: *const u8, kind: *const u8, n: usize) -> c_int;
    pub fn vk_sweep(e: *mut VkEngine, t: *const VkTransformSpec, seed_lo: u64, seed_hi_inclusive: u64, flags: u32, hits: *mut VkHit,
                    hit_cap: usize, out: *mut VkSweepStats) -> c_int;
    pub fn vk_generate(e: *mut VkEngine, t: *const VkTransformSpec, seed_lo: u64, seed_hi_inclusive: u64, flags: u32, privkeys: *mut u8,
                       h160_c: *mut u8, h160_u: *mut u8, valid: *mut u8, out: *mut VkSweepStats) -> c_int;
    pub fn vk_generate_lines(e: *mut VkEngine, t: *const VkTransformSpec, seed_lo: u64, seed_hi_inclusive: u64, transform_name: *const c_char,
                             slot_suffix: *const *const c_char, text: *mut u8, text_cap: usize, text_len: *mut usize,
                             out: *mut VkSweepStats) -> c_int;
    pub fn vk_generate_verbose(e: *mut VkEngine, t: *const VkTransformSpec, seed_lo: u64, seed_hi_inclusive: u64, transform_name: *const c_char,
                               slot_suffix: *const *const c_char, text: *mut u8, text_cap: usize, text_len: *mut usize,
                               out: *mut VkSweepStats) -> c_int;
    pub fn vk_generate_lines_strings(e: *mut VkEngine, t: *const VkTransformSpec, data: *const u8, offsets: *const u32, n: usize,
                                     label_data: *const u8, label_offsets: *const u32, transform_name: *const c_char, slot_suffix: *const *const c_char, text: *mut u8, text_cap: usize,
                                     text_len: *mut usize, out: *mut VkSweepStats) -> c_int;
    pub fn vk_record_column_stride(column: c_int) -> usize;
    pub fn vk_generate_columns(e: *mut VkEngine, t: *const VkTransformSpec, seed_lo: u64, seed_hi_inclusive: u64,
                               slot_suffix: *const *const c_char, cols: *mut VkRecordColumns, out: *mut VkSweepStats) -> c_int;
    pub fn vk_derive_columns(e: *mut VkEngine, keys: *const u8, n: usize, cols: *mut VkRecordColumns) -> c_int;
    pub fn vk_derive_batch(e: *mut VkEngine, keys: *const u8, n: usize, normalised: *mut u8, h160_c: *mut u8, h160_u: *mut u8) -> c_int;
    pub fn vk_derive_batch_pub(e: *mut VkEngine, keys: *const u8, n: usize, normalised: *mut u8, h160_c: *mut u8, h160_u: *mut u8,
                               pub_xy: *mut u8) -> c_int;
    pub fn vk_transform_batch(e: *mut VkEngine, t: *const VkTransformSpec, seeds: *const u64, n: usize, keys_out: *mut u8,
                              valid: *mut u8) -> c_int;
    pub fn vk_bip39_wordlist_set(e: *mut VkEngine, words: *const u8) -> c_int;
    pub fn vk_multibit_passphrase_set(e: *mut VkEngine, passphrase: *const u8, len: usize) -> c_int;
    pub fn vk_bitimage_keys(e: *mut VkEngine, entropy: *const u8, n_files: usize, pass_data: *const u8, pass_offsets: *const u32,
                            n_pass: usize, path: *const u32, depth: u32, count: u32, keys_out: *mut u8, valid: *mut u8) -> c_int;
    pub fn vk_transform_strings(e: *mut VkEngine, t: *const VkTransformSpec, data: *const u8, offsets: *const u32, n: usize,
                                keys_out: *mut u8, valid: *mut u8) -> c_int;
    pub fn vk_scan_keys(e: *mut VkEngine, keys: *const u8, n: usize, flags: u32, hits: *mut VkHit, hit_cap: usize,
                        out: *mut VkSweepStats) -> c_int;
    pub fn vk_scan_strings(e: *mut VkEngine, t: *const VkTransformSpec, data: *const u8, offsets: *const u32, n: usize, flags: u32,
                           hits: *mut VkHit, hit_cap: usize, out: *mut VkSweepStats) -> c_int;
    pub fn vk_rescan_strings(e: *mut VkEngine, flags: u32, hits: *mut VkHit, hit_cap: usize, out: *mut VkSweepStats) -> c_int;
    pub fn vk_search(e: *mut VkEngine, t: *const VkTransformSpec, seed_lo: u64, seed_hi_inclusive: u64, target: *const VkSearchTarget,
                     out: *mut VkSearchResult) -> c_int;
    pub fn vk_slots_per_input(t: *const VkTransformSpec) -> u32;
    pub fn vk_slots_per_string(t: *const VkTransformSpec) -> u32;
}
