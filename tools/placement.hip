// placement.hip -- round 5 diagnostic: where do the workgroups of a grid of (CUs x w) blocks of 256 lanes land, and do they run at
// the same time?  Every wave records its HW_ID (SE / CU / SIMD / wave slot), its XCC_ID and the constant-rate timestamps around a
// fixed amount of VALU work; the host prints how many waves shared each SIMD and how the time spans overlap.  (The sweep kernel
// launches exactly such a grid -- CUs x 2 persistent workgroups -- and the micro-benchmarks of tools/ubench*.hip read "w waves per
// SIMD" off it.)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/placement tools/placement.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
typedef uint32_t u32;
typedef uint64_t u64;
struct rec { u32 hw, xcc; u64 t0, t1, c0, c1; };
__global__ void __launch_bounds__(256) k(rec *out, int iters, u32 seed)
{
    u32 x = threadIdx.x * 2654435761u + seed + blockIdx.x;
    const u64 t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        asm volatile("v_alignbit_b32 %0, %0, %0, 7\n\tv_add_u32 %0, 0x428a2f98, %0\n\tv_alignbit_b32 %0, %0, %0, 11\n\tv_add_u32 %0, 0x71374491, %0\n\t"
                     "v_alignbit_b32 %0, %0, %0, 7\n\tv_add_u32 %0, 0x428a2f98, %0\n\tv_alignbit_b32 %0, %0, %0, 11\n\tv_add_u32 %0, 0x71374491, %0" : "+v"(x));
    }
    const u64 t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
    if (x == 0x12345678u) out[0].hw = x;
    if ((threadIdx.x & 63) == 0) {
        rec r;
        r.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        r.t0 = t0; r.t1 = t1; r.c0 = c0; r.c1 = c1;
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = r;
    }
}
int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    rec *d;
    (void)hipMalloc(&d, sizeof(rec) * cus * 8 * 4);
    for (int w : {1, 2, 3, 4}) {
        const int nb = cus * w;
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, 10, 1u);
        (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, iters, 7u);
        (void)hipDeviceSynchronize();
        std::vector<rec> h(nb * 4);
        (void)hipMemcpy(h.data(), d, sizeof(rec) * nb * 4, hipMemcpyDeviceToHost);
        std::map<u32, int> per_simd, per_cu;
        u64 tmin = ~0ull, tmax = 0;
        double dur = 0;
        for (auto &r : h) {
            const u32 simd = (r.hw >> 4) & 3, cu = (r.hw >> 8) & 15, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7, xcc = r.xcc & 15;
            const u32 cukey = (xcc << 16) | (se << 8) | (sh << 4) | cu;
            per_cu[cukey]++;
            per_simd[(cukey << 2) | simd]++;
            tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1);
            dur += (double)(r.t1 - r.t0);
        }
        std::map<int, int> hist_simd, hist_cu;
        for (auto &kv : per_simd) hist_simd[kv.second]++;
        for (auto &kv : per_cu) hist_cu[kv.second]++;
        // how many waves are running at the middle of the launch
        const u64 mid = (tmin + tmax) / 2;
        int live = 0;
        for (auto &r : h) live += (r.t0 <= mid && r.t1 >= mid);
        printf("grid %4d x 256 (w = %d): CUs used %zu, SIMDs used %zu;  waves per SIMD histogram:", nb, w, per_cu.size(), per_simd.size());
        for (auto &kv : hist_simd) printf("  %d waves: %d SIMDs", kv.first, kv.second);
        printf(";  waves per CU:");
        for (auto &kv : hist_cu) printf("  %d: %d CUs", kv.first, kv.second);
        printf(";  launch span %.1f us, mean wave span %.1f us, waves alive at mid-launch %d of %d;  cycles per instruction of wave 0: %.2f\n",
               (tmax - tmin) / 100.0, dur / h.size() / 100.0, live, nb * 4, (double)(h[0].c1 - h[0].c0) / (8.0 * iters));
    }
    return 0;
}
