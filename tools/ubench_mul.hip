// ubench_mul.hip -- one question: is a 256 x 256 -> 512-bit product cheaper on gfx950's double-precision FMA
// than on v_mad_u64_u32?
//
//   mad   the shipped product: 8 x 32-bit limbs, product scanning in even / odd columns (fe_mul_cols + merge_eo of
//         vuke_amd/csrc/vk_field_gen.cuh: 64 v_mad_u64_u32 + carry counters + one merge chain) -> 16 limbs
//   fma   5 x 52-bit limbs held as doubles (260 bits).  A 52 x 52 product has 104 bits; with the rounding mode at
//         round-toward-zero,  hi = fma(a, b, 2^104)  carries its top 52 bits in the mantissa and
//         lo = fma(a, b, (2^104 + 2^52) - hi)  the low 52 (Emmart, Zheng, Weems: "Faster modular exponentiation using double
//         precision floating point arithmetic on the GPU", ARITH 2018).  The mantissas are summed column by column as 64-bit
//         integers (bit-casts), the exponent patterns leave in one subtraction per column, then one carry pass and the way back
//         to doubles (what a field layer built on this form pays before the next product).  25 limb products x (2 FMA + 1 FP
//         subtraction + 2 integer additions) against 64 multiply-adds + 64 carry / merge instructions.
//
// Both kernels run a dependent chain of products (the result's low half feeds the next product), two waves per SIMD on every
// SIMD like the sweep kernel; clock and cycles are measured in-kernel (s_memtime / s_memrealtime).  A few lanes are checked
// against a host big-integer product.  `ubench_mul <mad|fma> <seconds>` loops one variant for power sampling
// (tools/probe_mul_power.sh).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mul tools/ubench_mul.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <chrono>
#include <vector>
typedef uint32_t u32;
typedef uint64_t u64;
#include "../vuke_amd/csrc/vk_field.cuh"

#define PRODUCTS_PER_ITER 4

__global__ void __launch_bounds__(256, 2) k_mad(u32 *out, u64 *clk, u32 seed, int iters)
{
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    fe a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a.v[i] = (t + 1) * 2654435761u + i * 40503u + seed;
        b.v[i] = ((t + 7) ^ seed) * 2246822519u + i * 977u + 1;
    }
    const u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    u32 tt[16];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < PRODUCTS_PER_ITER; u++) {
            u64 acc[15];
            const u32 top = fe_mul_cols(acc, a, b);
            merge_eo(tt, acc, top);
#pragma unroll
            for (int i = 0; i < 8; i++) a.v[i] = tt[i] ^ tt[8 + i]; // the next product depends on all 16 limbs of this one (8 VALU: not counted as product work)
        }
    }
    const u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 16; i++) out[(u64)t * 16 + i] = tt[i];
    if (t == 0) {
        clk[0] = c1 - c0;
        clk[1] = r1 - r0;
    }
}

__device__ __forceinline__ u64 as_u64(double d) { return __builtin_bit_cast(u64, d); }
__device__ __forceinline__ double as_f64(u64 v) { return __builtin_bit_cast(double, v); }

// r[10] (52-bit limbs, as integers) = a[5] x b[5] (doubles holding integers below 2^52); rounding mode must be RTZ
__device__ __forceinline__ void mul_fma(u64 r[10], const double a[5], const double b[5])
{
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    u64 col[10];
#pragma unroll
    for (int k = 0; k < 10; k++) col[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double hi = __builtin_fma(a[i], b[j], C1);
            const double lo = __builtin_fma(a[i], b[j], C2 - hi);
            col[i + j] += as_u64(lo);
            col[i + j + 1] += as_u64(hi);
        }
    }
    // column k received n_lo(k) low halves (pattern of 2^52) and n_hi(k) high halves (pattern of 2^104): one subtraction each
    const u64 P52 = 0x4330000000000000ull, P104 = 0x4670000000000000ull;
    u64 carry = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        const int nlo = k < 9 ? (k < 5 ? k + 1 : 9 - k) : 0;
        const int nhi = k > 0 ? (k - 1 < 5 ? k : 10 - k) : 0;
        const u64 v = col[k] - (u64)nlo * P52 - (u64)nhi * P104 + carry;
        r[k] = v & ((1ull << 52) - 1);
        carry = v >> 52;
    }
}

__global__ void __launch_bounds__(256, 2) k_fma(u64 *out, u64 *clk, u32 seed, int iters)
{
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3"); // double-precision rounding: toward zero
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    double a[5], b[5];
    const u64 M = (1ull << 52) - 1, P52 = 0x4330000000000000ull;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const u64 x = ((u64)((t + 1) * 2654435761u + i * 40503u + seed) << 21) ^ ((t + 3) * 2246822519u + i);
        const u64 y = ((u64)(((t + 7) ^ seed) * 2246822519u + i * 977u + 1) << 20) ^ ((t + 5) * 2654435761u + i);
        a[i] = as_f64(P52 | (x & M)) - 0x1p52;
        b[i] = as_f64(P52 | (y & M)) - 0x1p52;
    }
    const u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    u64 r[10];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < PRODUCTS_PER_ITER; u++) {
            mul_fma(r, a, b);
#pragma unroll
            for (int i = 0; i < 5; i++) a[i] = as_f64(P52 | (r[i] ^ r[5 + i])) - 0x1p52; // back to doubles: 1 OR (+ the chain's XOR) + 1 FP subtraction per limb
        }
    }
    const u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 10; i++) out[(u64)t * 10 + i] = r[i];
    if (t == 0) {
        clk[0] = c1 - c0;
        clk[1] = r1 - r0;
    }
}

// ---- host check: the same chains with 128-bit integers ---------------------------------------------------------------------
typedef unsigned __int128 u128;
static void big_mul(std::vector<u32> &r, const std::vector<u32> &a, const std::vector<u32> &b)
{
    r.assign(a.size() + b.size(), 0);
    for (size_t i = 0; i < a.size(); i++) {
        u64 c = 0;
        for (size_t j = 0; j < b.size(); j++) {
            u64 v = (u64)a[i] * b[j] + r[i + j] + c;
            r[i + j] = (u32)v;
            c = v >> 32;
        }
        r[i + b.size()] = (u32)c;
    }
}
static bool check_mad(const u32 *out, u32 t, u32 seed, int products)
{
    std::vector<u32> a(8), b(8), r;
    for (int i = 0; i < 8; i++) {
        a[i] = (t + 1) * 2654435761u + i * 40503u + seed;
        b[i] = ((t + 7) ^ seed) * 2246822519u + i * 977u + 1;
    }
    for (int p = 0; p < products; p++) {
        big_mul(r, a, b);
        for (int i = 0; i < 8; i++) a[i] = r[i] ^ r[8 + i];
    }
    return memcmp(r.data(), out + (u64)t * 16, 64) == 0;
}
static bool check_fma(const u64 *out, u32 t, u32 seed, int products)
{
    const u64 M = (1ull << 52) - 1;
    u64 a[5], b[5], r[10];
    for (int i = 0; i < 5; i++) {
        a[i] = (((u64)((t + 1) * 2654435761u + i * 40503u + seed) << 21) ^ ((t + 3) * 2246822519u + i)) & M;
        b[i] = (((u64)(((t + 7) ^ seed) * 2246822519u + i * 977u + 1) << 20) ^ ((t + 5) * 2654435761u + i)) & M;
    }
    for (int p = 0; p < products; p++) {
        u128 col[10] = {};
        for (int i = 0; i < 5; i++)
            for (int j = 0; j < 5; j++) col[i + j] += (u128)a[i] * b[j];
        u128 c = 0;
        for (int k = 0; k < 10; k++) {
            u128 v = col[k] + c;
            r[k] = (u64)v & M;
            c = v >> 52;
        }
        for (int i = 0; i < 5; i++) a[i] = r[i] ^ r[5 + i];
    }
    return memcmp(r, out + (u64)t * 10, 80) == 0;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * 2; // 256 lanes = 1 wave per SIMD: 2 blocks per CU = 2 waves per SIMD
    const u32 lanes = (u32)blocks * 256, seed = 12345;
    u32 *d_mad;
    u64 *d_fma, *d_clk;
    hipMalloc(&d_mad, (size_t)lanes * 64);
    hipMalloc(&d_fma, (size_t)lanes * 80);
    hipMalloc(&d_clk, 16);
    if (argc >= 3) { // power mode: one variant in a loop for N seconds
        const double secs = atof(argv[2]);
        const bool fma = !strcmp(argv[1], "fma");
        const double t0 = now();
        int n = 0;
        while (now() - t0 < secs) {
            if (fma) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, d_fma, d_clk, seed, 20000);
            else hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(256), 0, 0, d_mad, d_clk, seed, 20000);
            hipDeviceSynchronize();
            n++;
        }
        printf("%s: %d launches of %d products per lane in %.1f s = %.1f G products/s\n", argv[1], n, 20000 * PRODUCTS_PER_ITER, now() - t0,
               (double)n * 20000 * PRODUCTS_PER_ITER * lanes / (now() - t0) / 1e9);
        return 0;
    }
    printf("device %s  CUs %d, %d lanes (2 waves per SIMD)\n", prop.gcnArchName, cus, lanes);
    for (int pass = 0; pass < 2; pass++) {
        const int iters = pass == 0 ? 3 : 5000;
        for (int which = 0; which < 2; which++) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            if (which == 0) hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(256), 0, 0, d_mad, d_clk, seed, iters);
            else hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, d_fma, d_clk, seed, iters);
            hipEventRecord(e1, 0);
            hipError_t err = hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            u64 clk[2];
            hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost);
            const int products = iters * PRODUCTS_PER_ITER;
            if (pass == 0) { // correctness on a few lanes
                bool ok = true;
                if (which == 0) {
                    std::vector<u32> h((size_t)lanes * 16);
                    hipMemcpy(h.data(), d_mad, h.size() * 4, hipMemcpyDeviceToHost);
                    for (u32 t : {0u, 1u, 63u, 64u, 12345u, lanes - 1}) ok = ok && check_mad(h.data(), t, seed, products);
                } else {
                    std::vector<u64> h((size_t)lanes * 10);
                    hipMemcpy(h.data(), d_fma, h.size() * 8, hipMemcpyDeviceToHost);
                    for (u32 t : {0u, 1u, 63u, 64u, 12345u, lanes - 1}) ok = ok && check_fma(h.data(), t, seed, products);
                }
                printf("%s: %s against the host big-integer product (%d chained products, 6 lanes) [%s]\n", which ? "fma" : "mad", ok ? "EQUAL" : "DIFFERENT", products,
                       hipGetErrorString(err));
                continue;
            }
            const double cyc = (double)clk[0] / products / 2.0; // two waves per SIMD share it: cycles of SIMD time per wave-product
            const double ghz = (double)clk[0] / ((double)clk[1] / 100e6) / 1e9;
            printf("%s: %8.1f SIMD-cycles per wave-product (2 waves/SIMD), %6.2f G products/s chip-wide, clock %.2f GHz, %.2f ms\n", which ? "fma" : "mad", cyc,
                   (double)products * lanes / (ms * 1e-3) / 1e9, ghz, ms);
        }
    }
    return 0;
}
