/* vuke_hip_test_hooks.h -- test hooks of the HIP engine.  NOT part of the drop-in boundary (include/vuke_hip.h): these
 * functions exist only in libvuke_hip_hooks.so, the build of the same sources with -DVK_TEST_HOOKS that tests/ and tools/ load
 * when they need to look inside an engine (table entries, field arithmetic, the intermediate state of the key-derivation
 * chains, the launch policy of the chain ring).  The shipped libvuke_hip.so exports none of them
 * (tests/test_abi.py::test_shipped_library_exports_no_test_hooks). */
#ifndef VUKE_HIP_TEST_HOOKS_H
#define VUKE_HIP_TEST_HOOKS_H
#include "vuke_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: (d << (window_bits*w)) * G table entry as 64 bytes X||Y big-endian. */
int vk_debug_table_entry(vk_engine *e, uint32_t window, uint32_t digit, uint8_t xy[64]);
/* Test hook: n bytes at `offset` of the engine's device scratch slot `slot` (inputs, outputs and chain state of the last
 * batch call; layout is internal).  Used by the GPU tests to look at intermediate values of the key-derivation chains. */
int vk_debug_scratch(vk_engine *e, int slot, size_t offset, uint8_t *out, size_t n);
/* Test hook: the multibit chain of the following vk_transform_strings calls stops after `stages` of its five kernels
 * (1 prepare, 2 + first stretch, 3 + repacking, 4 + second stretch; 0 = run to the end, the default), leaving the chain state
 * in scratch slot 7 for vk_debug_scratch.  No keys are produced by a stopped chain. */
int vk_debug_chain_stop(vk_engine *e, int stages);

/* Test hook: field multiply / inverse on the device, big-endian 32-byte operands. */
int vk_debug_field(vk_engine *e, int op /* 0 mul, 1 sqr, 2 inv, 3 add, 4 sub */, const uint8_t *a, const uint8_t *b,
                   size_t n, uint8_t *out);

/* Test hook: the chain ring of the following vk_transform_strings / vk_scan_strings calls advances a group by `chunk_iters`
 * iterations per visit, keeps at most `waves_per_simd` worker waves resident per SIMD and starts `workers` worker waves (which may
 * exceed the number of groups: the waiting paths of the ring).  0 = the library's own choice for that knob. */
int vk_debug_ring_knobs(vk_engine *e, uint32_t chunk_iters, uint32_t waves_per_simd, uint32_t workers);
/* Test hook: exclusive prefix sum (mod 2^32) of n u32 through the library's single-pass scan kernel (vk_scan.cuh), the one that
 * places the compacted text lines / column pages of generate mode. */
int vk_debug_scan(vk_engine *e, const uint32_t *in, size_t n, uint32_t *out);

/* Test hook: allocation-failure injection.  The hooks build replaces operator new for the library's own allocations; the k-th next
 * one made by the calling thread inside the library (0 = the very next) fails once -- std::bad_alloc, or NULL from the nothrow form.
 * after < 0 disarms.  Returns the number of allocations the thread made through the library since the previous call.  The boundary's
 * promise under test: such a failure comes back as VK_E_NOMEM with vk_last_error set, never as an exception or a terminate. */
long long vk_debug_fail_alloc(long long after);
/* Test hook: the last sweep launch of the engine: out[0] = the kernel form (2 = two waves per SIMD, 256-lane workgroups; 3 = three
 * waves, compressed keys only; 4 = three waves as one 768-lane workgroup per CU, uncompressed keys hashed), out[1] = workgroups,
 * out[2] = lanes per workgroup, out[3] = inputs per lane per tile.  bench.py's name of the kernel it prices is held to this. */
int vk_debug_last_launch(vk_engine *e, uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
