// vk_field_prims_host.h -- TEST INFRASTRUCTURE: the field layer's instruction primitives for ONE lane on the host.
// tests/host_emu/host_field_emu.cpp compiles vuke_amd/csrc/vk_field.cuh with VK_FIELD_PRIMS_H / VK_FIELD_GEN_H pointing here and
// at vk_field_gen_host.h (tools/gen_field.py's second backend: every generated asm statement as calls into the emulator below),
// and tests/test_host_field.py holds the arithmetic to python integers on a machine without a GPU.  Nothing in the device build
// sees this directory.
#pragma once
// ---- VK_HOST_EMU: the field layer compiled for the host, one lane, every generated asm statement replaced by calls into the
// instruction emulator below (vk_field_gen_host.h, written by tools/gen_field.py next to the device header).  TEST
// INFRASTRUCTURE: tests/test_host_field.py builds it with g++ and holds the arithmetic to python integers on a machine without
// a GPU.  Nothing in the device build sees this branch.
#define VK_DEV static inline
#define __device__
#define __noinline__ __attribute__((noinline))
#define __constant__
#define __builtin_amdgcn_ballot_w64(x) ((x) ? 1ull : 0ull)
// semantics of the ten instructions the generator emits, for one lane; carries are the lane's bit 0 of a 64-bit mask
static inline void emu_mad(u64 &d, u64 &c, u32 a, u32 b, u64 add)
{
    const unsigned __int128 t = (unsigned __int128)a * b + add;
    d = (u64)t;
    c = (u64)(t >> 64) & 1u;
}
static inline void emu_add_co(u32 &d, u64 &c, u32 x, u32 y) { const u64 t = (u64)x + y; d = (u32)t; c = t >> 32; }
// low half of a pinned (counter, 0) pair (tools/gen_field.py, VK_GEN_PIN): the host is little-endian
static inline u32 &emu_lo32(u64 &v) { return *reinterpret_cast<u32 *>(&v); }
static inline void emu_addc_co(u32 &d, u64 &co, u32 x, u32 y, u64 ci) { const u64 t = (u64)x + y + (ci & 1u); d = (u32)t; co = t >> 32; }
static inline void emu_sub_co(u32 &d, u64 &c, u32 x, u32 y) { c = x < y; d = x - y; }
static inline void emu_subb_co(u32 &d, u64 &co, u32 x, u32 y, u64 ci) { const u64 s = (u64)y + (ci & 1u); const u64 b = (u64)x < s; d = (u32)((u64)x - s); co = b; }
static inline void emu_cndmask(u32 &d, u32 x, u32 y, u64 m) { d = (m & 1u) ? y : x; }
static inline void emu_add(u32 &d, u32 x, u32 y) { d = x + y; }
static inline void emu_mul_u24(u32 &d, u32 x, u32 y) { d = (x & 0xFFFFFFu) * (y & 0xFFFFFFu); }
static inline void emu_lshlrev(u32 &d, u32 sh, u32 x) { d = x << (sh & 31u); }
static inline void emu_alignbit(u32 &d, u32 hi, u32 lo, u32 sh) { d = (u32)((((u64)hi << 32) | lo) >> (sh & 31u)); }
static inline void emu_s_or(u64 &d, u64 x, u64 y) { d = x | y; }

typedef unsigned long vk_carry;
VK_DEV void mad_nc(u64 &acc, u32 a, u32 b) { u64 c; emu_mad(acc, c, a, b, acc); }
VK_DEV void mad_cy(u64 &acc, u32 &cy, u32 a, u32 b) { u64 c; emu_mad(acc, c, a, b, acc); emu_addc_co(cy, c, 0u, cy, c); }
VK_DEV u32 add_co(vk_carry &co, u32 x, u32 y) { u32 r; u64 c; emu_add_co(r, c, x, y); co = c; return r; }
VK_DEV u32 addc_co(vk_carry &co, u32 x, u32 y, vk_carry ci) { u32 r; u64 c; emu_addc_co(r, c, x, y, ci); co = c; return r; }
VK_DEV u32 addc(u32 x, u32 y, vk_carry ci) { u32 r; u64 c; emu_addc_co(r, c, x, y, ci); return r; }
VK_DEV u64 mul_wide_s(u32 a, u32 s) { u64 r, c; emu_mad(r, c, a, s, 0ull); return r; }
VK_DEV bool lane_in_mask(vk_carry mask) { return mask & 1ull; }
