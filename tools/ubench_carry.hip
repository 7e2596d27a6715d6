// ubench_carry.hip -- round 5: is the 8.75-cycle single-wave issue interval of v_mad_u64_u32 / v_addc_co_u32
// (profiles/archive/r03_ubench_valu_rates.txt) a property of the instruction class or of the carry register being reused?
// Patterns (inline asm, independent accumulators, all CUs, 1 / 2 / 3 waves per SIMD):
//   mad_vcc     8 multiply-adds, every one writes its carry to vcc                      (round 3's measurement: WAW on vcc)
//   mad_8sgpr   8 multiply-adds, each writes its carry to its own SGPR pair             (no reuse within 8 instructions)
//   mul2        the field multiplier's shipped pattern: mad(vcc) mad(s0) addc(vcc) addc(s0) ...   (two carry registers, reuse distance 2)
//   mul4        the same work with four carry registers: mad x4, addc x4                   (reuse distance 4)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_carry tools/ubench_carry.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32;
typedef uint64_t u64;
enum { MAD_VCC, MAD_8SGPR, MUL2, MUL4, NPAT };
static const char *NAMES[NPAT] = {"mad_vcc   (8 mads, all carries -> vcc)", "mad_8sgpr (8 mads, 8 carry pairs)", "mul2      (mad mad addc addc, 2 carry regs)",
                                  "mul4      (mad x4 addc x4, 4 carry regs)"};
static const int INSTR_PER_ITER[NPAT] = {8, 8, 8, 8};

template <int PAT>
__global__ void __launch_bounds__(256) k(u64 *out, u32 seed, int iters)
{
    u32 t = blockIdx.x * 256 + threadIdx.x;
    u32 a[8], b[8], cy[8];
    u64 c[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        a[j] = t * 2654435761u + j * 40503u + seed;
        b[j] = (t ^ seed) * 2246822519u + j + 1;
        c[j] = ((u64)a[j] << 32) | b[j];
        cy[j] = j;
    }
    u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (PAT == MAD_VCC) {
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %10, %1\n\tv_mad_u64_u32 %2, vcc, %8, %11, %2\n\tv_mad_u64_u32 %3, vcc, %8, %12, %3\n\t"
                             "v_mad_u64_u32 %4, vcc, %8, %13, %4\n\tv_mad_u64_u32 %5, vcc, %8, %14, %5\n\tv_mad_u64_u32 %6, vcc, %8, %15, %6\n\tv_mad_u64_u32 %7, vcc, %8, %16, %7"
                             : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
                             : "v"(a[u]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
                             : "vcc");
            } else if (PAT == MAD_8SGPR) {
                asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n\tv_mad_u64_u32 %1, s[22:23], %8, %10, %1\n\tv_mad_u64_u32 %2, s[24:25], %8, %11, %2\n\t"
                             "v_mad_u64_u32 %3, s[26:27], %8, %12, %3\n\tv_mad_u64_u32 %4, s[28:29], %8, %13, %4\n\tv_mad_u64_u32 %5, s[30:31], %8, %14, %5\n\t"
                             "v_mad_u64_u32 %6, s[32:33], %8, %15, %6\n\tv_mad_u64_u32 %7, s[34:35], %8, %16, %7"
                             : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
                             : "v"(a[u]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
                             : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
            } else if (PAT == MUL2) {
                // two columns (even -> vcc, odd -> s[20:21]), as tools/gen_field.py emits them; twice per block = 8 instructions
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, s[20:21], %4, %6, %1\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc\n\tv_addc_co_u32 %3, s[20:21], 0, %3, s[20:21]\n\t"
                             "v_mad_u64_u32 %0, vcc, %4, %7, %0\n\tv_mad_u64_u32 %1, s[20:21], %4, %8, %1\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc\n\tv_addc_co_u32 %3, s[20:21], 0, %3, s[20:21]"
                             : "+v"(c[0]), "+v"(c[1]), "+v"(cy[0]), "+v"(cy[1])
                             : "v"(a[u]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])
                             : "vcc", "s20", "s21");
            } else {
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, s[20:21], %8, %10, %1\n\tv_mad_u64_u32 %2, s[22:23], %8, %11, %2\n\tv_mad_u64_u32 %3, s[24:25], %8, %12, %3\n\t"
                             "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\tv_addc_co_u32 %5, s[20:21], 0, %5, s[20:21]\n\tv_addc_co_u32 %6, s[22:23], 0, %6, s[22:23]\n\t"
                             "v_addc_co_u32 %7, s[24:25], 0, %7, s[24:25]"
                             : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(cy[0]), "+v"(cy[1]), "+v"(cy[2]), "+v"(cy[3])
                             : "v"(a[u]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])
                             : "vcc", "s20", "s21", "s22", "s23", "s24", "s25");
            }
        }
    }
    u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 r = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) r ^= (u32)c[j] ^ (u32)(c[j] >> 32) ^ cy[j];
    if (r == 0x12345678u) out[3] = r;
    if (t == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

template <int PAT>
static void run(int waves_per_simd, int cus, u64 *out)
{
    const int iters = 4000;
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // exactly waves_per_simd workgroups on EVERY CU: each asks for a 1/waves_per_simd share of the CU's 160 KB of LDS, so no CU can take
    // one more (round 6: without it a 3-per-CU grid landed 4 / 2 on some CUs and the wall-clock figure read 4/3 of the truth)
    const size_t lds = (size_t)(160 * 1024) / waves_per_simd - 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<PAT>, grid, block, lds, 0, out, 1u, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PAT>, grid, block, lds, 0, out, 7u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    u64 h[2];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    double instr = (double)iters * 8 * INSTR_PER_ITER[PAT];
    printf("%-46s waves/SIMD=%d  one wave: %5.2f cycles between its instructions;  SIMD: %5.2f cycles per instruction (wall)  clk %.2f GHz\n", NAMES[PAT], waves_per_simd,
           (double)h[0] / instr, (ms * 1e-3) * ghz * 1e9 / (instr * waves_per_simd), ghz);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("device %s  CUs %d\n", p.gcnArchName, cus);
    u64 *out;
    hipMalloc(&out, 64);
    for (int w : {1, 2, 3, 4}) {
        run<MAD_VCC>(w, cus, out);
        run<MAD_8SGPR>(w, cus, out);
        run<MUL2>(w, cus, out);
        run<MUL4>(w, cus, out);
        printf("\n");
    }
    return 0;
}
