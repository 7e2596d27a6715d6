// hip_shims.h -- TEST INFRASTRUCTURE: HIP's kernel syntax and the handful of gfx950 builtins the cold kernels use, for compiling
// them for the host one lane at a time (host_chain_emu.cpp, host_records_emu.cpp).  Wave-level operations (__shfl) have no
// one-lane meaning: kernels that use them compile against a stub and are not called by the harnesses.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define VK_FIELD_PRIMS_H "vk_field_prims_host.h"
#define VK_FIELD_GEN_H "vk_field_gen_host.h"

// ---- HIP kernel syntax, one lane
struct emu_dim3 {
    unsigned x = 0, y = 0, z = 0;
};
static emu_dim3 threadIdx, blockIdx, blockDim;
#define __global__
#define __launch_bounds__(...)
#define __forceinline__ inline
using std::min;

// ---- gfx950 builtins used by the chain code
static inline uint32_t emu_alignbit32(uint32_t hi, uint32_t lo, uint32_t n) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (n & 31u)); }
static inline uint32_t emu_bitop3(uint32_t a, uint32_t b, uint32_t c, uint32_t tbl)
{
    uint32_t r = 0; // bit idx of tbl = the function's value on (a, b, c) = the three bits of idx
    for (uint32_t idx = 0; idx < 8; idx++)
        if ((tbl >> idx) & 1u) r |= ((idx & 4u) ? a : ~a) & ((idx & 2u) ? b : ~b) & ((idx & 1u) ? c : ~c);
    return r;
}
#define __builtin_amdgcn_alignbit(hi, lo, n) emu_alignbit32((hi), (lo), (n))
#define __builtin_amdgcn_alignbyte(hi, lo, n) emu_alignbit32((hi), (lo), 8u * ((n) & 3u))
#define __builtin_amdgcn_bitop3_b32(a, b, c, t) emu_bitop3((a), (b), (c), (t))
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
template <class T>
static inline T emu_fetch_add(T *p, T v)
{
    const T o = *p;
    *p = o + v;
    return o;
}
#define __hip_atomic_fetch_add(p, v, order, scope) emu_fetch_add((p), (v))
#define __host__
#define __shfl(v, lane, width) (v)
