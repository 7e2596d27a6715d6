// vk_scan.cuh -- exclusive prefix sum of u32 lengths, single pass (decoupled look-back, Merrill & Garland): the offsets of the
// compacted text lines / Parquet PLAIN pages of generate mode (vk_lines.cuh, vk_records.cuh).  Rounds 1-3 called
// hipcub::DeviceScan for this (~25 trampoline kernels in the code object for one 30-line kernel's worth of work).
//
// A workgroup takes the next tile from a ticket counter (so every predecessor of a running tile is running or done: the look-back
// cannot wait for a workgroup that has not started), sums its 2048 items, publishes (1, aggregate), walks back over its
// predecessors' words until one carries (2, inclusive prefix), publishes its own inclusive prefix, writes its outputs.  Flag and
// value share one 64-bit word, handed over with release / acquire at agent scope (the L2 is per XCD).  Sums wrap mod 2^32.
#pragma once
#include "vk_field.cuh" // u32, VK_DEV

#define VK_SCAN_ITEMS 8u
#define VK_SCAN_TILE (256u * VK_SCAN_ITEMS)

// state[0] = ticket counter, state[1 + t] = (flag << 32) | value of tile t; all zero before the launch
__global__ void __launch_bounds__(256) vk_scan_kernel(const u32 *__restrict__ in, u32 *__restrict__ out, u32 n, unsigned long long *state)
{
    __shared__ u32 s_tile, s_wave[4], s_prefix;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_tile = (u32)atomicAdd(&state[0], 1ull);
    __syncthreads();
    const u32 tile = s_tile;
    const u32 base = tile * VK_SCAN_TILE + tid * VK_SCAN_ITEMS;
    u32 v[VK_SCAN_ITEMS], sum = 0;
#pragma unroll
    for (u32 i = 0; i < VK_SCAN_ITEMS; i++) {
        v[i] = base + i < n ? in[base + i] : 0u;
        sum += v[i];
    }
    u32 incl = sum; // inclusive scan of the lane sums over the wave
#pragma unroll
    for (u32 o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    u32 wave_off = 0, total = 0;
#pragma unroll
    for (u32 w = 0; w < 4; w++) {
        if (w < wave) wave_off += s_wave[w];
        total += s_wave[w];
    }
    if (tid == 0) {
        u32 prefix = 0;
        if (tile == 0) {
            __hip_atomic_store(&state[1], (2ull << 32) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(&state[1 + tile], (1ull << 32) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            u32 t = tile - 1; // tile 0 always ends the walk: it publishes flag 2
            for (;;) {
                const unsigned long long s = __hip_atomic_load(&state[1 + t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                const u32 flag = (u32)(s >> 32);
                if (flag == 0) {
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                prefix += (u32)s;
                if (flag == 2) break;
                t--;
            }
            __hip_atomic_store(&state[1 + tile], (2ull << 32) | (u32)(prefix + total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_prefix = prefix;
    }
    __syncthreads();
    u32 run = s_prefix + wave_off + incl - sum;
#pragma unroll
    for (u32 i = 0; i < VK_SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
}
