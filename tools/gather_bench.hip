// gather_bench.hip -- how many random 64-byte table reads per second HBM + TLB sustain on this part (the k*G table access
// pattern of the sweep kernel, with no arithmetic in the way).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/gather_bench tools/gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32;
typedef uint64_t u64;

__global__ void __launch_bounds__(256) gather(const uint4 *__restrict__ t, u64 entries_mask, u32 iters, u32 inflight, u64 *out)
{
    u64 x = ((u64)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u32 i = 0; i < iters; i++) {
        uint4 v[8][4];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if ((u32)k < inflight) {
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                const uint4 *p = t + ((x >> 20) & entries_mask) * 4;
                v[k][0] = p[0]; v[k][1] = p[1]; v[k][2] = p[2]; v[k][3] = p[3];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((u32)k < inflight)
#pragma unroll
                for (int q = 0; q < 4; q++) { acc.x ^= v[k][q].x; acc.y += v[k][q].y; acc.z ^= v[k][q].z; acc.w += v[k][q].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

// the same number of 64-byte entries, fetched by groups of four lanes: lane l of a group reads quarter (l & 3) of the entry of
// lane (group base + round), so one load instruction touches 16 whole lines instead of 64 lines a quarter each
__global__ void __launch_bounds__(256) gather_coop(const uint4 *__restrict__ t, u64 entries_mask, u32 iters, u64 *out)
{
    u64 x = ((u64)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const u32 lane = threadIdx.x & 63, q = lane & 3;
    for (u32 i = 0; i < iters; i++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const u64 mine = (x >> 20) & entries_mask;
        uint4 v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const u32 src = (lane & ~3u) | (u32)r;
            const u32 lo = __shfl((u32)mine, src, 64), hi = __shfl((u32)(mine >> 32), src, 64);
            const u64 idx = ((u64)hi << 32) | lo;
            v[r] = t[idx * 4 + q];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) { acc.x ^= v[r].x; acc.y += v[r].y; acc.z ^= v[r].z; acc.w += v[r].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 16.0;
    u64 entries = 1;
    while (entries * 2 * 64 <= (u64)(gb * 1e9)) entries *= 2;
    uint4 *t;
    u64 *out;
    if (hipMalloc(&t, entries * 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 8);
    hipMemset(t, 1, entries * 64);
    for (u32 waves_per_simd : {2u, 4u, 8u})
        for (u32 inflight : {1u, 2u, 4u, 8u}) {
            const u32 grid = 256 * waves_per_simd, iters = 2048 / inflight;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(gather, dim3(grid), dim3(256), 0, 0, t, entries - 1, 16u, inflight, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(gather, dim3(grid), dim3(256), 0, 0, t, entries - 1, iters, inflight, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)grid * 256 * iters * inflight;
            printf("table %.1f GB  waves/SIMD %u  loads in flight per lane %u: %.2f G gathers/s  %.2f TB/s\n", entries * 64 / 1e9, waves_per_simd, inflight,
                   n / (ms * 1e-3) / 1e9, n * 64 / (ms * 1e-3) / 1e12);
        }
    for (u32 waves_per_simd : {2u, 4u, 8u}) {
        const u32 grid = 256 * waves_per_simd, iters = 2048;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(gather_coop, dim3(grid), dim3(256), 0, 0, t, entries - 1, 16u, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(gather_coop, dim3(grid), dim3(256), 0, 0, t, entries - 1, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)grid * 256 * iters;
        printf("table %.1f GB  waves/SIMD %u  four lanes per entry: %.2f G gathers/s  %.2f TB/s\n", entries * 64 / 1e9, waves_per_simd, n / (ms * 1e-3) / 1e9,
               n * 64 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
