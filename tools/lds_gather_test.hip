// lds_gather_test.hip -- checks the landing layout of global_load_lds_dwordx4 for the quad-cooperative table read
// (four rounds; lane L loads quarter L&3 of the entry owned by quad lane r; owner reads 64 contiguous bytes) and times it.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32;
typedef uint64_t u64;
#define ROUND_STRIDE 65 /* uint4 per round: 64 lanes + 16 bytes of padding */

__device__ __forceinline__ void issue(uint4 *tile /* wave-uniform */, const uint4 *table, u32 entry)
{
    const u32 q = threadIdx.x & 3u;
    const u32 e0 = __builtin_amdgcn_update_dpp(0u, entry, 0x00, 0xF, 0xF, false);
    const u32 e1 = __builtin_amdgcn_update_dpp(0u, entry, 0x55, 0xF, 0xF, false);
    const u32 e2 = __builtin_amdgcn_update_dpp(0u, entry, 0xAA, 0xF, 0xF, false);
    const u32 e3 = __builtin_amdgcn_update_dpp(0u, entry, 0xFF, 0xF, 0xF, false);
    typedef const void __attribute__((address_space(1))) *gp;
    typedef void __attribute__((address_space(3))) *lp;
    __builtin_amdgcn_global_load_lds((gp)(table + ((u64)e0 << 2) + q), (lp)(tile + 0 * ROUND_STRIDE), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gp)(table + ((u64)e1 << 2) + q), (lp)(tile + 1 * ROUND_STRIDE), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gp)(table + ((u64)e2 << 2) + q), (lp)(tile + 2 * ROUND_STRIDE), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gp)(table + ((u64)e3 << 2) + q), (lp)(tile + 3 * ROUND_STRIDE), 16, 0, 0);
}

__global__ void __launch_bounds__(256) k(const uint4 *table, u64 mask, u32 iters, u64 *out)
{
    __shared__ uint4 tiles[4 * 4 * ROUND_STRIDE];
    uint4 *tile = tiles + (threadIdx.x >> 6) * 4 * ROUND_STRIDE;
    const u32 lane = threadIdx.x & 63u;
    u64 x = ((u64)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 777;
    u64 bad = 0;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u32 i = 0; i < iters; i++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const u32 entry = (u32)((x >> 20) & mask);
        issue(tile, table, entry);
        __builtin_amdgcn_s_waitcnt(0); // everything (vmcnt, lgkmcnt)
        const uint4 *mine = tile + (lane & 3u) * ROUND_STRIDE + (lane & ~3u);
        const uint4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
        if (out[1]) { // verification mode
            const uint4 *p = table + ((u64)entry << 2);
            const uint4 ra = p[0], rb = p[1], rc = p[2], rd = p[3];
            bad += (a.x != ra.x) | (a.w != ra.w) | (b.y != rb.y) | (c.z != rc.z) | (d.x != rd.x) | (d.w != rd.w);
        }
        acc.x ^= a.x ^ b.y; acc.y += c.z; acc.z ^= d.w;
        __builtin_amdgcn_s_waitcnt(0);
    }
    if (bad) atomicAdd((unsigned long long *)&out[2], (unsigned long long)bad);
    if ((acc.x ^ acc.y ^ acc.z) == 0x1234567u) out[3] = 1;
}

__global__ void fill(uint4 *t, u64 n)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) t[i] = make_uint4((u32)i, (u32)(i >> 32) ^ 0x5a5a, (u32)(i * 2654435761u), (u32)(i * 40503u + 7));
}

int main()
{
    const u64 entries = 1ull << 28; // 16 GB
    uint4 *t;
    u64 *out;
    if (hipMalloc(&t, entries * 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 64);
    hipMemset(out, 0, 64);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, t, entries * 4);
    u64 one = 1;
    hipMemcpy(out + 1, &one, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, t, entries - 1, 64u, out);
    hipDeviceSynchronize();
    u64 h[4];
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("verification: %llu mismatching lanes (of %u)\n", (unsigned long long)h[2], 512u * 256 * 64);
    hipMemset(out, 0, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, t, entries - 1, 2048u, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("one gather in flight per lane, wait after each: %.2f G gathers/s\n", 2048.0 * 256 * 2048 / (ms * 1e-3) / 1e9);
    return h[2] ? 1 : 0;
}
