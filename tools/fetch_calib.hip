// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE on the access patterns of the sweep kernel (MI355X_MICROARCH.md:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel below
// reads an exactly known number of bytes from a table far larger than L2 + Infinity Cache; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -o pmc -- tools/fetch_calib
// and divide (tools/fetch_calib.py).  Patterns:
//   calib_stream      16 B per lane, fully coalesced (the batch-inversion workspace reads)
//   calib_gather64    every lane reads its own random 64-byte entry with four 16-byte loads (per-lane table read, the probe)
//   calib_gather_lds  the sweep's table read: quads fetch whole random 64-byte entries with global_load_lds_dwordx4
//   calib_gather32    every lane reads one random 32-byte slot (the target-table probe)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32;
typedef uint64_t u64;
typedef const void __attribute__((address_space(1))) *gptr;
typedef void __attribute__((address_space(3))) *lptr;

__device__ __forceinline__ u64 lcg(u64 &x)
{
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    return x >> 20;
}

__global__ void __launch_bounds__(256) calib_stream(const uint4 *__restrict__ t, u64 n16, u32 iters, u64 *out)
{
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x, nth = (u64)gridDim.x * 256;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u32 i = 0; i < iters; i++) {
        const uint4 v = t[(tid + (u64)i * nth) % n16];
        acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

__global__ void __launch_bounds__(256) calib_gather64(const uint4 *__restrict__ t, u64 mask, u32 iters, u64 *out)
{
    u64 x = ((u64)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u32 i = 0; i < iters; i++) {
        const uint4 *p = t + (lcg(x) & mask) * 4;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc.x ^= a.x ^ b.x; acc.y += c.y + d.y; acc.z ^= a.z ^ d.z; acc.w += b.w + c.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

__global__ void __launch_bounds__(256) calib_gather32(const uint4 *__restrict__ t, u64 mask, u32 iters, u64 *out)
{
    u64 x = ((u64)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 777;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u32 i = 0; i < iters; i++) {
        const uint4 *p = t + (lcg(x) & (2 * mask + 1)) * 2;
        const uint4 a = p[0], b = p[1];
        acc.x ^= a.x ^ b.x; acc.y += a.y + b.y; acc.z ^= a.z ^ b.z; acc.w += a.w + b.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

#define ROUND 65
__global__ void __launch_bounds__(256) calib_gather_lds(const uint4 *__restrict__ t, u64 mask, u32 iters, u64 *out)
{
    __shared__ uint4 tile[4 * 4 * ROUND];
    uint4 *slot = tile + (threadIdx.x >> 6) * 4 * ROUND;
    u64 x = ((u64)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 999;
    const u32 lane = threadIdx.x & 63, q = lane & 3;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u32 i = 0; i < iters; i++) {
        const u32 entry = (u32)(lcg(x) & mask);
        const u32 e0 = __builtin_amdgcn_update_dpp(0u, entry, 0x00, 0xF, 0xF, false), e1 = __builtin_amdgcn_update_dpp(0u, entry, 0x55, 0xF, 0xF, false);
        const u32 e2 = __builtin_amdgcn_update_dpp(0u, entry, 0xAA, 0xF, 0xF, false), e3 = __builtin_amdgcn_update_dpp(0u, entry, 0xFF, 0xF, 0xF, false);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_global_load_lds((gptr)(t + ((u64)e0 << 2) + q), (lptr)(slot + 0 * ROUND), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(t + ((u64)e1 << 2) + q), (lptr)(slot + 1 * ROUND), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(t + ((u64)e2 << 2) + q), (lptr)(slot + 2 * ROUND), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(t + ((u64)e3 << 2) + q), (lptr)(slot + 3 * ROUND), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4 *m = slot + q * ROUND + (lane & ~3u);
        const uint4 a = m[0], d = m[3];
        acc.x ^= a.x; acc.y += d.y; acc.z ^= a.z; acc.w += d.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 32.0;
    u64 entries = 1;
    while (entries * 2 * 64 <= (u64)(gb * 1e9) && entries * 2 <= (1ull << 32)) entries *= 2;
    uint4 *t;
    u64 *out;
    if (hipMalloc(&t, entries * 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 8);
    hipMemset(t, 1, entries * 64);
    hipDeviceSynchronize();
    const u32 grid = 256 * 8, iters = 1024;
    const u64 lanes = (u64)grid * 256;
    hipLaunchKernelGGL(calib_stream, dim3(grid), dim3(256), 0, 0, t, entries * 4, iters, out);
    hipDeviceSynchronize();
    printf("calib_stream bytes %llu\n", (unsigned long long)(lanes * iters * 16));
    hipLaunchKernelGGL(calib_gather64, dim3(grid), dim3(256), 0, 0, t, entries - 1, iters, out);
    hipDeviceSynchronize();
    printf("calib_gather64 bytes %llu\n", (unsigned long long)(lanes * iters * 64));
    hipLaunchKernelGGL(calib_gather32, dim3(grid), dim3(256), 0, 0, t, entries - 1, iters, out);
    hipDeviceSynchronize();
    printf("calib_gather32 bytes %llu\n", (unsigned long long)(lanes * iters * 32));
    hipLaunchKernelGGL(calib_gather_lds, dim3(grid), dim3(256), 0, 0, t, entries - 1, iters, out);
    hipDeviceSynchronize();
    printf("calib_gather_lds bytes %llu\n", (unsigned long long)(lanes * iters * 64));
    printf("table_bytes %llu\n", (unsigned long long)(entries * 64));
    return 0;
}
