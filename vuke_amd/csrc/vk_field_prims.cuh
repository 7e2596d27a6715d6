// vk_field_prims.cuh -- the instruction-level primitives of the field layer (vk_field.cuh) as gfx950 inline asm.
#pragma once
#include <hip/hip_runtime.h>

#define VK_DEV __device__ __forceinline__

// acc = a * b + acc; the sum is known not to overflow 64 bits
VK_DEV void mad_nc(u64 &acc, u32 a, u32 b)
{
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
// acc = a * b + acc (mod 2^64); cy += carry out
VK_DEV void mad_cy(u64 &acc, u32 &cy, u32 a, u32 b)
{
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(cy) : "v"(a), "v"(b) : "vcc");
}
// carry chains with the carry in an SGPR pair, so the compiler can schedule freely between links
typedef unsigned long vk_carry;
VK_DEV u32 add_co(vk_carry &co, u32 x, u32 y)
{
    u32 r;
    asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(r), "=s"(co) : "v"(x), "v"(y));
    return r;
}
VK_DEV u32 addc_co(vk_carry &co, u32 x, u32 y, vk_carry ci)
{
    u32 r;
    asm("v_addc_co_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(co) : "v"(x), "v"(y), "s"(ci));
    return r;
}
VK_DEV u32 addc(u32 x, u32 y, vk_carry ci) // last link: carry out is known to be zero
{
    u32 r;
    vk_carry co;
    asm("v_addc_co_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(co) : "v"(x), "v"(y), "s"(ci));
    return r;
}

// a * s (s in an SGPR), full 64-bit product
VK_DEV u64 mul_wide_s(u32 a, u32 s)
{
    u64 r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a), "s"(s) : "vcc");
    return r;
}

VK_DEV bool lane_in_mask(vk_carry mask)
{
    u32 lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return (mask >> lane) & 1ull;
}
