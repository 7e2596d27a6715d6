// ubench_block.hip -- issue cost of the VALU instruction classes the sweep's mix is priced with, per number of resident waves per
// SIMD (1, 2, 3, 4, 8), measured the way the shipped kernels present them to the SIMD: EIGHT independent instructions per asm block
// (no s_nop between them; tools/ubench.hip puts every instruction into its own asm statement, the compiler separates those, and
// with three waves on a SIMD that artefact reads as 3.1 cycles for a plain add where two and four waves reach 2.4 -- round 6).
// Every CU holds exactly `waves` workgroups of 256 lanes (one wave per SIMD each): the launch asks for a 1/waves share of the LDS.
// Output lines have the format of tools/ubench.hip, so tools/ubench_rates.py reads either.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_block tools/ubench_block.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32;
typedef uint64_t u64;

enum { ADD, XOR, ADD3, BFI, ALIGNBIT, ADDC, LSHLADD, XAD, PERM, MULLO, MULHI, MAD64, MOV, BITOP3, CMP, LSHLADD64, NOPS };
static const char *NAMES[NOPS] = {"v_add_u32", "v_xor_b32", "v_add3_u32", "v_bfi_b32", "v_alignbit_b32", "v_addc_co_u32", "v_lshl_add_u32", "v_xad_u32",
                                  "v_perm_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mov_b32", "v_bitop3_b32", "v_cmp_lt_u32", "v_lshl_add_u64"};

// eight instructions `pre dst, dst, src post`, dst = %0..%7, src = %8..%15
#define L2(pre, post, d, s) pre " %" #d ", %" #d ", %" #s post "\n\t"
#define B2(pre, post) L2(pre, post, 0, 8) L2(pre, post, 1, 9) L2(pre, post, 2, 10) L2(pre, post, 3, 11) L2(pre, post, 4, 12) L2(pre, post, 5, 13) L2(pre, post, 6, 14) L2(pre, post, 7, 15)
// `pre dst, dst, src, src post`
#define L3(pre, post, d, s) pre " %" #d ", %" #d ", %" #s ", %" #s post "\n\t"
#define B3(pre, post) L3(pre, post, 0, 8) L3(pre, post, 1, 9) L3(pre, post, 2, 10) L3(pre, post, 3, 11) L3(pre, post, 4, 12) L3(pre, post, 5, 13) L3(pre, post, 6, 14) L3(pre, post, 7, 15)
#define AOPS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define BOPS "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
#define COPS "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])

template <int OP>
__global__ void __launch_bounds__(256) k(u64 *out, u32 seed, int iters)
{
    u32 t = blockIdx.x * 256 + threadIdx.x;
    u32 a[8], b[8];
    u64 c[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        a[j] = t * 2654435761u + j * 40503u + seed;
        b[j] = (t ^ seed) * 2246822519u + j + 1;
        c[j] = ((u64)a[j] << 32) | b[j];
    }
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[0]), "v"(b[0]) : "vcc");
    u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == ADD) asm volatile(B2("v_add_u32", "") : AOPS : BOPS);
            if (OP == XOR) asm volatile(B2("v_xor_b32", "") : AOPS : BOPS);
            if (OP == MULLO) asm volatile(B2("v_mul_lo_u32", "") : AOPS : BOPS);
            if (OP == MULHI) asm volatile(B2("v_mul_hi_u32", "") : AOPS : BOPS);
            if (OP == ALIGNBIT) asm volatile(B2("v_alignbit_b32", ", 7") : AOPS : BOPS);
            if (OP == ADD3) asm volatile(B3("v_add3_u32", "") : AOPS : BOPS);
            if (OP == BFI) asm volatile(B3("v_bfi_b32", "") : AOPS : BOPS);
            if (OP == XAD) asm volatile(B3("v_xad_u32", "") : AOPS : BOPS);
            if (OP == PERM) asm volatile(B3("v_perm_b32", "") : AOPS : BOPS);
            if (OP == BITOP3) asm volatile(B3("v_bitop3_b32", " bitop3:0x96") : AOPS : BOPS);
            if (OP == ADDC)
                asm volatile("v_addc_co_u32 %0, vcc, %0, %8, vcc\n\tv_addc_co_u32 %1, vcc, %1, %9, vcc\n\tv_addc_co_u32 %2, vcc, %2, %10, vcc\n\tv_addc_co_u32 %3, vcc, %3, %11, vcc\n\t"
                             "v_addc_co_u32 %4, vcc, %4, %12, vcc\n\tv_addc_co_u32 %5, vcc, %5, %13, vcc\n\tv_addc_co_u32 %6, vcc, %6, %14, vcc\n\tv_addc_co_u32 %7, vcc, %7, %15, vcc"
                             : AOPS : BOPS : "vcc");
            if (OP == LSHLADD)
                asm volatile("v_lshl_add_u32 %0, %0, 3, %8\n\tv_lshl_add_u32 %1, %1, 3, %9\n\tv_lshl_add_u32 %2, %2, 3, %10\n\tv_lshl_add_u32 %3, %3, 3, %11\n\t"
                             "v_lshl_add_u32 %4, %4, 3, %12\n\tv_lshl_add_u32 %5, %5, 3, %13\n\tv_lshl_add_u32 %6, %6, 3, %14\n\tv_lshl_add_u32 %7, %7, 3, %15" : AOPS : BOPS);
            if (OP == MOV)
                asm volatile("v_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\tv_mov_b32 %2, %10\n\tv_mov_b32 %3, %11\n\tv_mov_b32 %4, %12\n\tv_mov_b32 %5, %13\n\tv_mov_b32 %6, %14\n\tv_mov_b32 %7, %15"
                             : AOPS : BOPS);
            if (OP == CMP)
                asm volatile("v_cmp_lt_u32 vcc, %0, %8\n\tv_cmp_lt_u32 vcc, %1, %9\n\tv_cmp_lt_u32 vcc, %2, %10\n\tv_cmp_lt_u32 vcc, %3, %11\n\t"
                             "v_cmp_lt_u32 vcc, %4, %12\n\tv_cmp_lt_u32 vcc, %5, %13\n\tv_cmp_lt_u32 vcc, %6, %14\n\tv_cmp_lt_u32 vcc, %7, %15" : AOPS : BOPS : "vcc");
            if (OP == MAD64)
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %16, %0\n\tv_mad_u64_u32 %1, vcc, %9, %17, %1\n\tv_mad_u64_u32 %2, vcc, %10, %18, %2\n\tv_mad_u64_u32 %3, vcc, %11, %19, %3\n\t"
                             "v_mad_u64_u32 %4, vcc, %12, %20, %4\n\tv_mad_u64_u32 %5, vcc, %13, %21, %5\n\tv_mad_u64_u32 %6, vcc, %14, %22, %6\n\tv_mad_u64_u32 %7, vcc, %15, %23, %7"
                             : COPS : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), BOPS : "vcc");
            if (OP == LSHLADD64)
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n\tv_lshl_add_u64 %1, %1, 0, %2\n\tv_lshl_add_u64 %2, %2, 0, %3\n\tv_lshl_add_u64 %3, %3, 0, %4\n\t"
                             "v_lshl_add_u64 %4, %4, 0, %5\n\tv_lshl_add_u64 %5, %5, 0, %6\n\tv_lshl_add_u64 %6, %6, 0, %7\n\tv_lshl_add_u64 %7, %7, 0, %0" : COPS);
        }
    }
    u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 r = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) r ^= a[j] ^ (u32)c[j] ^ (u32)(c[j] >> 32);
    if (r == 0x12345678u) out[3] = r; // keep everything live
    if (t == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

template <int OP>
static void run(int waves, int cus, u64 *out)
{
    const int iters = 4000;
    dim3 grid(cus * waves), block(256);
    const size_t lds = (size_t)(160 * 1024) / waves - 1024; // at most `waves` workgroups fit a CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, grid, block, lds, 0, out, 1u, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, grid, block, lds, 0, out, 7u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    u64 h[2];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    double per_wave = (double)iters * 64;
    printf("%-18s waves/SIMD=%d live=64  %6.2f cyc/wave-instr/SIMD (wall %5.2f)  %8.1f Tlane-ops/s (64-lane equiv)  clk %.2f GHz  (%.2f ms)\n", NAMES[OP], waves,
           (double)h[0] / (per_wave * waves), (ms * 1e-3) * ghz * 1e9 / (per_wave * waves), (double)grid.x * 256 * per_wave / (ms * 1e-3) / 1e12, ghz, ms);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}
template <int OP>
static void sweep(int cus, u64 *out)
{
    for (int w : {1, 2, 3, 4, 8}) run<OP>(w, cus, out);
    printf("\n");
}

int main()
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
    int cus = p.multiProcessorCount;
    printf("device %s  CUs %d  nominal clock %d kHz  (eight instructions per asm block)\n", p.gcnArchName, cus, p.clockRate);
    u64 *out;
    if (hipMalloc(&out, 64) != hipSuccess) return 1;
    sweep<ADD>(cus, out);
    sweep<XOR>(cus, out);
    sweep<ADD3>(cus, out);
    sweep<BFI>(cus, out);
    sweep<ALIGNBIT>(cus, out);
    sweep<ADDC>(cus, out);
    sweep<LSHLADD>(cus, out);
    sweep<XAD>(cus, out);
    sweep<PERM>(cus, out);
    sweep<MULLO>(cus, out);
    sweep<MULHI>(cus, out);
    sweep<MAD64>(cus, out);
    sweep<MOV>(cus, out);
    sweep<BITOP3>(cus, out);
    sweep<CMP>(cus, out);
    sweep<LSHLADD64>(cus, out);
    return 0;
}
