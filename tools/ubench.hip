// ubench.hip -- per-instruction VALU throughput on gfx950 for the ops the key-derivation
// path is made of.  Prints giga-lane-ops/s for the whole chip and cycles per wave-instruction
// per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32;
typedef uint64_t u64;
#define CHAINS 8

template <int OP>
__global__ void __launch_bounds__(256) k(u32 *out, u32 seed, int iters)
{
    u32 t = blockIdx.x * 256 + threadIdx.x;
    u32 a[CHAINS], b[CHAINS];
    u64 c[CHAINS];
    double d[CHAINS];
#pragma unroll
    for (int j = 0; j < CHAINS; j++) {
        a[j] = t * 2654435761u + j * 40503u + seed;
        b[j] = (t ^ seed) * 2246822519u + j + 1;
        c[j] = ((u64)a[j] << 32) | b[j];
        d[j] = 1.0 + 1e-9 * (double)(a[j] & 0xffff);
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int j = 0; j < CHAINS; j++) {
                if (OP == 0) a[j] = a[j] + b[j];
                if (OP == 1) a[j] = a[j] ^ (b[j] & a[j]) ;            // v_bfi/and+xor class
                if (OP == 2) a[j] = a[j] * b[j];                      // v_mul_lo_u32
                if (OP == 3) a[j] = __umulhi(a[j], b[j]) + 1u;        // v_mul_hi_u32 (+add)
                if (OP == 4) c[j] = (u64)(u32)c[j] * (u64)b[j] + c[j];// v_mad_u64_u32
                if (OP == 5) a[j] = __builtin_amdgcn_alignbit(a[j], a[j], 7) ^ b[j];
                if (OP == 6) a[j] = a[j] + b[j] + (u32)j;             // v_add3_u32
                if (OP == 7) a[j] = __builtin_amdgcn_alignbit(a[j], b[j], 13);
                if (OP == 8) a[j] = (a[j] & 0xffffffu) * (b[j] & 0xffffffu) + a[j]; // v_mad_u32_u24 class
                if (OP == 9) d[j] = __builtin_fma(d[j], 1.0000001, 1e-12);   // v_fma_f64
                if (OP == 10) c[j] = c[j] + (((u64)b[j] << 32) | a[j]);      // 64-bit add = add_co + addc
            }
        }
    }
    u32 r = 0;
#pragma unroll
    for (int j = 0; j < CHAINS; j++) r ^= a[j] ^ (u32)c[j] ^ (u32)(c[j] >> 32) ^ (u32)d[j];
    out[t] = r;
}

template <int OP>
static void run(const char *name, int blocks_per_cu, int cus, u32 *out)
{
    const int iters = 2000;
    dim3 grid(cus * blocks_per_cu), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 1u, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 7u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)grid.x * 256 * iters * 4.0 * CHAINS;
    double gops = ops / (ms * 1e-3) / 1e9;
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    double wave_instr = ops / 64.0;
    double simds = cus * 4.0;
    double cyc = (ms * 1e-3) * 2.4e9 / (wave_instr / simds);
    printf("%-28s waves/SIMD=%d  %9.1f Glane-ops/s   %.2f cyc/wave-instr/SIMD @2.4GHz  (%.3f ms)\n", name, blocks_per_cu, gops, cyc, ms);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("device %s  CUs %d  clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
    u32 *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    for (int occ : {1, 2, 4, 8}) {
        run<0>("v_add_u32", occ, cus, out);
        run<1>("and+xor (2 ops)", occ, cus, out);
        run<2>("v_mul_lo_u32", occ, cus, out);
        run<3>("v_mul_hi_u32 + add (2 ops)", occ, cus, out);
        run<4>("v_mad_u64_u32", occ, cus, out);
        run<5>("alignbit + xor (2 ops)", occ, cus, out);
        run<6>("v_add3_u32", occ, cus, out);
        run<7>("v_alignbit_b32", occ, cus, out);
        run<8>("mad_u32_u24 class (3 ops)", occ, cus, out);
        run<9>("v_fma_f64", occ, cus, out);
        run<10>("64-bit add (2 ops)", occ, cus, out);
        printf("\n");
    }
    return 0;
}
