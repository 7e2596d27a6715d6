//! Bindings to `libvuke_hip.so` (`include/vuke_hip.h`, ABI version 5) and, behind the `vuke-source` feature, the
//! `HipRangeSource` that plugs the engine into vuke's `Source` seam (`src/source/mod.rs:23-31` of the reference).
pub mod ffi;
#[cfg(feature = "vuke-source")]
pub mod hip_range;

use std::ffi::CStr;

/// What a negative return code of the C ABI means (`include/vuke_hip.h`, "Error convention"): errors are values, nothing unwinds
/// across the boundary -- the library catches `std::bad_alloc` and every other C++ exception at its `extern "C"` frames (ABI note of
/// round 6), so none of these calls can abort the Rust caller.  Mirrors the reference's `GpuError` (`src/gpu/error.rs:7-20`).
pub fn error_kind(rc: i32) -> &'static str {
    match rc {
        ffi::VK_OK => "ok",
        ffi::VK_E_NO_DEVICE => "no usable gfx950 device (the caller falls back to its CPU path, like src/main.rs:1582-1587)",
        ffi::VK_E_HIP => "HIP runtime error, or an unexpected exception inside the library",
        ffi::VK_E_INVALID_ARG => "invalid argument",
        ffi::VK_E_UNSUPPORTED => "not supported by this build",
        ffi::VK_E_NOMEM => "out of memory: a device allocation (table, workspace, hit buffer) or a host staging buffer did not fit",
        _ => "unknown error code",
    }
}

/// `Err(message)` for a negative `VK_E_*` return code, with the engine's `vk_last_error` text.
pub fn check(engine: *const ffi::VkEngine, rc: i32) -> Result<(), String> {
    if rc == ffi::VK_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::vk_last_error(engine)) }.to_string_lossy().into_owned();
    Err(format!("vuke_hip error {rc} ({}): {msg}", error_kind(rc)))
}

/// Owned engine handle: one per GPU, internally serialised (safe to share between threads).
pub struct Engine(pub *mut ffi::VkEngine);
unsafe impl Send for Engine {}
unsafe impl Sync for Engine {}

impl Engine {
    pub fn new(device: i32) -> Result<Self, String> {
        let mut e: *mut ffi::VkEngine = std::ptr::null_mut();
        check(std::ptr::null(), unsafe { ffi::vk_engine_create(device, &mut e) })?;
        Ok(Engine(e))
    }
    /// An engine with a chosen table geometry (`vk_engine_create_ex`): e.g. `layout: 1, window_bits: 26` = ten signed windows, 18 GB.
    pub fn with_opts(device: i32, opts: &ffi::VkEngineOpts) -> Result<Self, String> {
        let mut e: *mut ffi::VkEngine = std::ptr::null_mut();
        check(std::ptr::null(), unsafe { ffi::vk_engine_create_ex(device, opts, &mut e) })?;
        Ok(Engine(e))
    }
    pub fn last_warning(&self) -> String {
        let mut buf = vec![0u8; 512];
        let n = unsafe { ffi::vk_last_warning(self.0, buf.as_mut_ptr().cast(), buf.len()) };
        buf.truncate(n.min(511));
        String::from_utf8_lossy(&buf).into_owned()
    }
}

/// The value columns of generate mode's full record for `[lo, hi]` (`vk_generate_columns`): buffers sized by
/// `vk_record_column_stride`, truncated to what the device wrote.  Column `c` holds Parquet PLAIN bytes; `plain_to_arrow`
/// turns a string column into the (offsets, values) pair an Arrow `StringArray` adopts without copying characters twice.
pub struct RecordColumns {
    pub rows: u64,
    pub data: Vec<Vec<u8>>,
}

impl Engine {
    pub fn generate_columns(&self, spec: &ffi::VkTransformSpec, lo: u64, hi: u64, slot_suffix: &[*const std::os::raw::c_char]) -> Result<(RecordColumns, ffi::VkSweepStats), String> {
        let slots = unsafe { ffi::vk_slots_per_input(spec) } as u64;
        let items = ((hi - lo + 1) * slots) as usize;
        let mut data: Vec<Vec<u8>> = (0..ffi::VK_RC_COLUMNS).map(|c| vec![0u8; items * unsafe { ffi::vk_record_column_stride(c as i32) }]).collect();
        let mut rc = ffi::VkRecordColumns { data: [std::ptr::null_mut(); ffi::VK_RC_COLUMNS], cap: [0; ffi::VK_RC_COLUMNS], len: [0; ffi::VK_RC_COLUMNS], rows: 0 };
        for c in 0..ffi::VK_RC_COLUMNS {
            rc.data[c] = data[c].as_mut_ptr();
            rc.cap[c] = data[c].len();
        }
        let mut st = ffi::VkSweepStats::default();
        check(self.0, unsafe { ffi::vk_generate_columns(self.0, spec, lo, hi, slot_suffix.as_ptr(), &mut rc, &mut st) })?;
        for c in 0..ffi::VK_RC_COLUMNS {
            data[c].truncate(rc.len[c]);
        }
        Ok((RecordColumns { rows: rc.rows, data }, st))
    }
}

/// PLAIN BYTE_ARRAY column (4-byte little-endian length + characters per row) -> Arrow's (i32 offsets, value bytes).
pub fn plain_to_arrow(plain: &[u8], rows: usize) -> (Vec<i32>, Vec<u8>) {
    let mut offsets = Vec::with_capacity(rows + 1);
    let mut values = Vec::with_capacity(plain.len().saturating_sub(4 * rows));
    let mut at = 0usize;
    offsets.push(0i32);
    for _ in 0..rows {
        let len = u32::from_le_bytes([plain[at], plain[at + 1], plain[at + 2], plain[at + 3]]) as usize;
        values.extend_from_slice(&plain[at + 4..at + 4 + len]);
        at += 4 + len;
        offsets.push(values.len() as i32);
    }
    (offsets, values)
}

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { ffi::vk_engine_destroy(self.0) }
    }
}
