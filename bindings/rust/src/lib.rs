//! Bindings to `libvuke_hip.so` (`include/vuke_hip.h`, ABI version 3) and, behind the `vuke-source` feature, the
//! `HipRangeSource` that plugs the engine into vuke's `Source` seam (`src/source/mod.rs:23-31` of the reference).
pub mod ffi;
#[cfg(feature = "vuke-source")]
pub mod hip_range;

use std::ffi::CStr;

/// `Err(message)` for a negative `VK_E_*` return code, with the engine's `vk_last_error` text.
pub fn check(engine: *const ffi::VkEngine, rc: i32) -> Result<(), String> {
    if rc == ffi::VK_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::vk_last_error(engine)) }.to_string_lossy().into_owned();
    Err(format!("vuke_hip error {rc}: {msg}"))
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
    pub fn last_warning(&self) -> String {
        let mut buf = vec![0u8; 512];
        let n = unsafe { ffi::vk_last_warning(self.0, buf.as_mut_ptr().cast(), buf.len()) };
        buf.truncate(n.min(511));
        String::from_utf8_lossy(&buf).into_owned()
    }
}

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { ffi::vk_engine_destroy(self.0) }
    }
}
