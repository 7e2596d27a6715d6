/* abi_layout.c -- TEST INFRASTRUCTURE: prints sizeof / offsetof of every struct of include/vuke_hip.h as "Struct.field N" lines
 * (field "." = the struct's size).  tests/test_abi_layout.py compares them with the layout assertions in bindings/rust/src/ffi.rs
 * and with the ctypes mirrors in vuke_amd/_ffi.py.  Compiled as strict C11: the header is plain C. */
#include "vuke_hip.h"
#include <stddef.h>
#include <stdio.h>

#define S(T) printf(#T " . %zu\n", sizeof(T))
#define F(T, f) printf(#T " " #f " %zu\n", offsetof(T, f))

int main(void)
{
    S(vk_transform_spec); F(vk_transform_spec, kind); F(vk_transform_spec, variant); F(vk_transform_spec, endian); F(vk_transform_spec, chain_depth);
    S(vk_hit); F(vk_hit, seed); F(vk_hit, sub_index); F(vk_hit, addr_kind); F(vk_hit, privkey); F(vk_hit, h160); F(vk_hit, _pad);
    S(vk_sweep_stats); F(vk_sweep_stats, inputs); F(vk_sweep_stats, keys); F(vk_sweep_stats, matches); F(vk_sweep_stats, hits_dropped);
    F(vk_sweep_stats, digest); F(vk_sweep_stats, kernel_seconds); F(vk_sweep_stats, launches); F(vk_sweep_stats, inputs_skipped);
    F(vk_sweep_stats, shader_clock_ghz);
    S(vk_engine_opts); F(vk_engine_opts, window_bits); F(vk_engine_opts, batch); F(vk_engine_opts, blocks_per_cu); F(vk_engine_opts, layout);
    F(vk_engine_opts, stream); F(vk_engine_opts, algo); F(vk_engine_opts, _pad);
    S(vk_engine_info_t); F(vk_engine_info_t, device); F(vk_engine_info_t, cu_count); F(vk_engine_info_t, window_bits); F(vk_engine_info_t, windows);
    F(vk_engine_info_t, batch); F(vk_engine_info_t, blocks_per_cu); F(vk_engine_info_t, layout); F(vk_engine_info_t, _pad);
    F(vk_engine_info_t, table_bytes); F(vk_engine_info_t, table_build_seconds);
    F(vk_engine_info_t, runtime_init_seconds); F(vk_engine_info_t, alloc_seconds); F(vk_engine_info_t, table_wall_seconds);
    S(vk_search_target); F(vk_search_target, mode); F(vk_search_target, n_cascade); F(vk_search_target, key); F(vk_search_target, mask_bits);
    F(vk_search_target, _pad); F(vk_search_target, bits); F(vk_search_target, targets);
    S(vk_search_result); F(vk_search_result, found); F(vk_search_result, n_keys); F(vk_search_result, seed); F(vk_search_result, seeds_checked);
    F(vk_search_result, kernel_seconds); F(vk_search_result, keys); F(vk_search_result, variant); F(vk_search_result, index);
    S(vk_record_columns); F(vk_record_columns, data); F(vk_record_columns, cap); F(vk_record_columns, len); F(vk_record_columns, rows);
    printf("VK_ABI_VERSION . %d\n", VK_ABI_VERSION);
    return 0;
}
