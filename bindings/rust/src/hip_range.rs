//! `HipRangeSource`: vuke's `RangeSource` (`src/source/range.rs:14-79`) on the MI355X engine.  Written against the reference's
//! traits (`Source`, `Transform`, `Matcher`, `Output`, `KeyDeriver`, `ProcessStats`), which the embedding crate brings into scope;
//! enabled with the `vuke-source` feature.  The fused sweep replaces the closure body of `RangeSource::process`
//! (`Transform::apply_batch` -> `KeyDeriver::derive` -> `Matcher::check`); the rare hits are replayed through the reference's own
//! `KeyDeriver` + `Output::hit` on the CPU, generate mode in the compact format takes its text from `vk_generate_lines`.
use crate::ffi::*;
use crate::{check, Engine};
use anyhow::Result;
use std::ffi::CString;
use vuke::derive::KeyDeriver;
use vuke::matcher::Matcher;
use vuke::output::Output;
use vuke::source::{ProcessStats, Source};
use vuke::transform::Transform;

pub struct HipRangeSource {
    pub start: u64,
    pub end: u64,
    engine: Engine,
    /// one spec per transform of the run, from `TransformType::hip_spec()`
    specs: Vec<VkTransformSpec>,
}

impl HipRangeSource {
    pub fn new(start: u64, end: u64, device: i32, specs: Vec<VkTransformSpec>) -> Result<Self> {
        Ok(Self { start, end, engine: Engine::new(device).map_err(anyhow::Error::msg)?, specs })
    }
}

/// What `Transform::apply_batch` appends to the input's decimal string in the label of slot `sub`
/// (`src/transform/lcg.rs:54-66`, `xorshift.rs:43-47`, `sha256_chain.rs:67`).  Held slot by slot to the oracle-made
/// `tests/golden/labels.json` by `tests/test_labels.py`, which replays this function from its source text: keep its shape.
pub fn label_suffix(spec: &VkTransformSpec, sub: u32) -> String {
    const LCG: [&str; 4] = ["glibc", "minstd", "msvc", "borland"];
    // XorshiftVariant::name() (src/xorshift.rs:36-43), NOT the short forms from_str also accepts
    const XS: [&str; 4] = ["xorshift64", "xorshift128", "xorshift128+", "xoroshiro128**"];
    const CH: [&str; 4] = ["iterated", "indexed:be", "indexed:le", "counter"];
    let all = spec.variant == VK_VARIANT_ALL;
    match spec.kind {
        VK_T_LCG if all => format!(":{}{}", LCG[(sub & 3) as usize], if spec.endian != 0 { ":le" } else { "" }),
        VK_T_XORSHIFT if all => format!(":{}", XS[(sub & 3) as usize]),
        VK_T_SHA256_CHAIN => {
            let (nv, d) = (if all { 4 } else { 1 }, spec.chain_depth.max(1));
            let r = sub % (nv * d);
            let v = if all { r / d } else { spec.variant };
            format!("[{}:{}]", CH[(v & 3) as usize], r % d)
        }
        _ => String::new(),
    }
}

impl Source for HipRangeSource {
    fn process(&self, transforms: &[Box<dyn Transform>], matcher: Option<&Matcher>, output: &dyn Output) -> Result<ProcessStats> {
        let e = self.engine.0;
        let deriver = KeyDeriver::new();
        let mut stats = ProcessStats { inputs_processed: self.end - self.start + 1, ..Default::default() };
        if let Some(m) = matcher {
            // Matcher gains `pub fn hash160_targets(&self) -> (Vec<u8>, Vec<u8>)`: every line that is a mainnet P2PKH (kind 0) or
            // P2WPKH (kind 1) address, decoded; other lines can never equal a derived address.
            let (h160, kind) = m.hash160_targets();
            check(e, unsafe { vk_targets_set(e, h160.as_ptr(), kind.as_ptr(), kind.len()) }).map_err(anyhow::Error::msg)?;
        }
        for (spec, t) in self.specs.iter().zip(transforms) {
            let slots = unsafe { vk_slots_per_input(spec) };
            if let Some(m) = matcher {
                let mut hits = vec![unsafe { std::mem::zeroed::<VkHit>() }; 1 << 16];
                let mut st = VkSweepStats::default();
                loop {
                    check(e, unsafe { vk_sweep(e, spec, self.start, self.end, VK_F_COMPRESSED | VK_F_UNCOMPRESSED, hits.as_mut_ptr(), hits.len(), &mut st) })
                        .map_err(anyhow::Error::msg)?;
                    if st.hits_dropped == 0 {
                        break;
                    }
                    hits.resize(hits.len() + st.hits_dropped as usize + 1024, unsafe { std::mem::zeroed() });
                }
                stats.keys_generated += st.keys;
                stats.matches_found += st.matches;
                hits.truncate(st.matches as usize);
                hits.sort_by_key(|h| (h.seed, h.sub_index));
                for h in &hits {
                    let derived = deriver.derive(&h.privkey); // CPU, rare
                    let info = m.check(&derived).expect("a device hit matches on the CPU too");
                    output.hit(&format!("{}{}", h.seed, label_suffix(spec, h.sub_index)), t.name(), &derived, &info).ok();
                }
            } else {
                // generate mode, compact console format: the text comes from the device, 2^21 (input, slot) pairs per call
                let suffix: Vec<CString> = (0..slots).map(|s| CString::new(label_suffix(spec, s)).unwrap()).collect();
                let ptrs: Vec<*const std::os::raw::c_char> = suffix.iter().map(|s| s.as_ptr()).collect();
                let name = CString::new(t.name()).unwrap();
                let chunk = ((1u64 << 21) / slots as u64).max(1);
                let mut text = vec![0u8; (chunk * slots as u64 * 192) as usize];
                let mut lo = self.start;
                loop {
                    let hi = if self.end - lo < chunk - 1 { self.end } else { lo + chunk - 1 };
                    let (mut len, mut st) = (0usize, VkSweepStats::default());
                    check(e, unsafe { vk_generate_lines(e, spec, lo, hi, name.as_ptr(), ptrs.as_ptr(), text.as_mut_ptr(), text.len(), &mut len, &mut st) })
                        .map_err(anyhow::Error::msg)?;
                    stats.keys_generated += st.keys;
                    output.write_text(&text[..len])?; // Output gains `fn write_text(&self, &[u8]) -> Result<()>` next to key()
                    if hi == self.end {
                        break;
                    }
                    lo = hi + 1;
                }
            }
        }
        if matcher.is_some() {
            check(e, unsafe { vk_targets_set(e, std::ptr::null(), std::ptr::null(), 0) }).map_err(anyhow::Error::msg)?;
        }
        Ok(stats)
    }
}
