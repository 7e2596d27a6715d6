// ubench_chain.hip -- round 5 (the asm bodies below were expanded by a throw-away script).  Do the SERIAL sections of the sweep
// kernel (MT19937's init_genrand: 404 dependent steps of shift / xor / multiply-add; the SHA-256 and RIPEMD-160 rounds) issue at the
// rate of their instruction classes when only two waves share a SIMD, or do they wait for their own results?  One lane runs 1, 2 or
// 4 INDEPENDENT chains, interleaved instruction by instruction; if a single chain is latency-bound, two chains per lane finish in
// less than twice the time.  Bodies are single asm statements on fixed registers (the compiler puts an s_nop between
// single-instruction asm statements).
//   mt      x = 1812433253 * (x ^ (x >> 30)) + 7          (v_lshrrev, v_xor, v_mad_u64_u32: the sweep's mt_init_step)
//   rot     x = (rotr(x, 7) ^ rotr(x, 18) ^ (x >> 3)) + k  (v_alignbit x2, v_lshrrev, v_bitop3, v_add: the hash rounds' classes)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_chain tools/ubench_chain.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32;
typedef uint64_t u64;

__global__ void __launch_bounds__(256) k_mt1(u64 *out, u32 seed, int iters)
{
    u32 t = blockIdx.x * 256 + threadIdx.x;
    u32 x0 = t * 2654435761u + seed;
    asm volatile("v_mov_b32 v20, %0\n\tv_add_u32 v22, 40503, %0\n\tv_add_u32 v24, 81006, %0\n\tv_add_u32 v26, 121509, %0\n\tv_mov_b32 v50, 0x6c078965" :: "v"(x0) : "v20", "v22", "v24", "v26", "v50");
    u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        asm volatile("v_lshrrev_b32 v40, 30, v20\n\tv_xor_b32 v40, v40, v20\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_xor_b32 v40, v40, v20\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_xor_b32 v40, v40, v20\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_xor_b32 v40, v40, v20\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_xor_b32 v40, v40, v20\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_xor_b32 v40, v40, v20\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_xor_b32 v40, v40, v20\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_xor_b32 v40, v40, v20\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
    }
    u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 r;
    asm volatile("v_xor_b32 %0, v20, v22\n\tv_xor_b32 %0, %0, v24\n\tv_xor_b32 %0, %0, v26" : "=v"(r));
    if (r == 0x12345678u) out[3] = r;
    if (t == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

__global__ void __launch_bounds__(256) k_mt2(u64 *out, u32 seed, int iters)
{
    u32 t = blockIdx.x * 256 + threadIdx.x;
    u32 x0 = t * 2654435761u + seed;
    asm volatile("v_mov_b32 v20, %0\n\tv_add_u32 v22, 40503, %0\n\tv_add_u32 v24, 81006, %0\n\tv_add_u32 v26, 121509, %0\n\tv_mov_b32 v50, 0x6c078965" :: "v"(x0) : "v20", "v22", "v24", "v26", "v50");
    u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        asm volatile("v_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
    }
    u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 r;
    asm volatile("v_xor_b32 %0, v20, v22\n\tv_xor_b32 %0, %0, v24\n\tv_xor_b32 %0, %0, v26" : "=v"(r));
    if (r == 0x12345678u) out[3] = r;
    if (t == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

__global__ void __launch_bounds__(256) k_mt4(u64 *out, u32 seed, int iters)
{
    u32 t = blockIdx.x * 256 + threadIdx.x;
    u32 x0 = t * 2654435761u + seed;
    asm volatile("v_mov_b32 v20, %0\n\tv_add_u32 v22, 40503, %0\n\tv_add_u32 v24, 81006, %0\n\tv_add_u32 v26, 121509, %0\n\tv_mov_b32 v50, 0x6c078965" :: "v"(x0) : "v20", "v22", "v24", "v26", "v50");
    u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        asm volatile("v_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_lshrrev_b32 v42, 30, v24\n\tv_lshrrev_b32 v43, 30, v26\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_xor_b32 v42, v42, v24\n\tv_xor_b32 v43, v43, v26\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_mad_u64_u32 v[24:25], vcc, v42, v50, 7\n\tv_mad_u64_u32 v[26:27], vcc, v43, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_lshrrev_b32 v42, 30, v24\n\tv_lshrrev_b32 v43, 30, v26\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_xor_b32 v42, v42, v24\n\tv_xor_b32 v43, v43, v26\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_mad_u64_u32 v[24:25], vcc, v42, v50, 7\n\tv_mad_u64_u32 v[26:27], vcc, v43, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_lshrrev_b32 v42, 30, v24\n\tv_lshrrev_b32 v43, 30, v26\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_xor_b32 v42, v42, v24\n\tv_xor_b32 v43, v43, v26\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_mad_u64_u32 v[24:25], vcc, v42, v50, 7\n\tv_mad_u64_u32 v[26:27], vcc, v43, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_lshrrev_b32 v42, 30, v24\n\tv_lshrrev_b32 v43, 30, v26\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_xor_b32 v42, v42, v24\n\tv_xor_b32 v43, v43, v26\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_mad_u64_u32 v[24:25], vcc, v42, v50, 7\n\tv_mad_u64_u32 v[26:27], vcc, v43, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_lshrrev_b32 v42, 30, v24\n\tv_lshrrev_b32 v43, 30, v26\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_xor_b32 v42, v42, v24\n\tv_xor_b32 v43, v43, v26\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_mad_u64_u32 v[24:25], vcc, v42, v50, 7\n\tv_mad_u64_u32 v[26:27], vcc, v43, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_lshrrev_b32 v42, 30, v24\n\tv_lshrrev_b32 v43, 30, v26\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_xor_b32 v42, v42, v24\n\tv_xor_b32 v43, v43, v26\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_mad_u64_u32 v[24:25], vcc, v42, v50, 7\n\tv_mad_u64_u32 v[26:27], vcc, v43, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_lshrrev_b32 v42, 30, v24\n\tv_lshrrev_b32 v43, 30, v26\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_xor_b32 v42, v42, v24\n\tv_xor_b32 v43, v43, v26\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_mad_u64_u32 v[24:25], vcc, v42, v50, 7\n\tv_mad_u64_u32 v[26:27], vcc, v43, v50, 7\n\tv_lshrrev_b32 v40, 30, v20\n\tv_lshrrev_b32 v41, 30, v22\n\tv_lshrrev_b32 v42, 30, v24\n\tv_lshrrev_b32 v43, 30, v26\n\tv_xor_b32 v40, v40, v20\n\tv_xor_b32 v41, v41, v22\n\tv_xor_b32 v42, v42, v24\n\tv_xor_b32 v43, v43, v26\n\tv_mad_u64_u32 v[20:21], vcc, v40, v50, 7\n\tv_mad_u64_u32 v[22:23], vcc, v41, v50, 7\n\tv_mad_u64_u32 v[24:25], vcc, v42, v50, 7\n\tv_mad_u64_u32 v[26:27], vcc, v43, v50, 7" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
    }
    u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 r;
    asm volatile("v_xor_b32 %0, v20, v22\n\tv_xor_b32 %0, %0, v24\n\tv_xor_b32 %0, %0, v26" : "=v"(r));
    if (r == 0x12345678u) out[3] = r;
    if (t == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

__global__ void __launch_bounds__(256) k_rot1(u64 *out, u32 seed, int iters)
{
    u32 t = blockIdx.x * 256 + threadIdx.x;
    u32 x0 = t * 2654435761u + seed;
    asm volatile("v_mov_b32 v20, %0\n\tv_add_u32 v22, 40503, %0\n\tv_add_u32 v24, 81006, %0\n\tv_add_u32 v26, 121509, %0\n\tv_mov_b32 v50, 0x6c078965" :: "v"(x0) : "v20", "v22", "v24", "v26", "v50");
    u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        asm volatile("v_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
    }
    u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 r;
    asm volatile("v_xor_b32 %0, v20, v22\n\tv_xor_b32 %0, %0, v24\n\tv_xor_b32 %0, %0, v26" : "=v"(r));
    if (r == 0x12345678u) out[3] = r;
    if (t == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

__global__ void __launch_bounds__(256) k_rot2(u64 *out, u32 seed, int iters)
{
    u32 t = blockIdx.x * 256 + threadIdx.x;
    u32 x0 = t * 2654435761u + seed;
    asm volatile("v_mov_b32 v20, %0\n\tv_add_u32 v22, 40503, %0\n\tv_add_u32 v24, 81006, %0\n\tv_add_u32 v26, 121509, %0\n\tv_mov_b32 v50, 0x6c078965" :: "v"(x0) : "v20", "v22", "v24", "v26", "v50");
    u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        asm volatile("v_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
    }
    u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 r;
    asm volatile("v_xor_b32 %0, v20, v22\n\tv_xor_b32 %0, %0, v24\n\tv_xor_b32 %0, %0, v26" : "=v"(r));
    if (r == 0x12345678u) out[3] = r;
    if (t == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

__global__ void __launch_bounds__(256) k_rot4(u64 *out, u32 seed, int iters)
{
    u32 t = blockIdx.x * 256 + threadIdx.x;
    u32 x0 = t * 2654435761u + seed;
    asm volatile("v_mov_b32 v20, %0\n\tv_add_u32 v22, 40503, %0\n\tv_add_u32 v24, 81006, %0\n\tv_add_u32 v26, 121509, %0\n\tv_mov_b32 v50, 0x6c078965" :: "v"(x0) : "v20", "v22", "v24", "v26", "v50");
    u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        asm volatile("v_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v42, v24, v24, 7\n\tv_alignbit_b32 v43, v26, v26, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_alignbit_b32 v46, v24, v24, 18\n\tv_alignbit_b32 v47, v26, v26, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_lshrrev_b32 v25, 3, v24\n\tv_lshrrev_b32 v27, 3, v26\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_bitop3_b32 v42, v42, v46, v25 bitop3:0x96\n\tv_bitop3_b32 v43, v43, v47, v27 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_add_u32 v24, 0x428a2f98, v42\n\tv_add_u32 v26, 0x428a2f98, v43\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v42, v24, v24, 7\n\tv_alignbit_b32 v43, v26, v26, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_alignbit_b32 v46, v24, v24, 18\n\tv_alignbit_b32 v47, v26, v26, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_lshrrev_b32 v25, 3, v24\n\tv_lshrrev_b32 v27, 3, v26\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_bitop3_b32 v42, v42, v46, v25 bitop3:0x96\n\tv_bitop3_b32 v43, v43, v47, v27 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_add_u32 v24, 0x428a2f98, v42\n\tv_add_u32 v26, 0x428a2f98, v43\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v42, v24, v24, 7\n\tv_alignbit_b32 v43, v26, v26, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_alignbit_b32 v46, v24, v24, 18\n\tv_alignbit_b32 v47, v26, v26, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_lshrrev_b32 v25, 3, v24\n\tv_lshrrev_b32 v27, 3, v26\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_bitop3_b32 v42, v42, v46, v25 bitop3:0x96\n\tv_bitop3_b32 v43, v43, v47, v27 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_add_u32 v24, 0x428a2f98, v42\n\tv_add_u32 v26, 0x428a2f98, v43\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v42, v24, v24, 7\n\tv_alignbit_b32 v43, v26, v26, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_alignbit_b32 v46, v24, v24, 18\n\tv_alignbit_b32 v47, v26, v26, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_lshrrev_b32 v25, 3, v24\n\tv_lshrrev_b32 v27, 3, v26\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_bitop3_b32 v42, v42, v46, v25 bitop3:0x96\n\tv_bitop3_b32 v43, v43, v47, v27 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_add_u32 v24, 0x428a2f98, v42\n\tv_add_u32 v26, 0x428a2f98, v43\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v42, v24, v24, 7\n\tv_alignbit_b32 v43, v26, v26, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_alignbit_b32 v46, v24, v24, 18\n\tv_alignbit_b32 v47, v26, v26, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_lshrrev_b32 v25, 3, v24\n\tv_lshrrev_b32 v27, 3, v26\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_bitop3_b32 v42, v42, v46, v25 bitop3:0x96\n\tv_bitop3_b32 v43, v43, v47, v27 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_add_u32 v24, 0x428a2f98, v42\n\tv_add_u32 v26, 0x428a2f98, v43\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v42, v24, v24, 7\n\tv_alignbit_b32 v43, v26, v26, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_alignbit_b32 v46, v24, v24, 18\n\tv_alignbit_b32 v47, v26, v26, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_lshrrev_b32 v25, 3, v24\n\tv_lshrrev_b32 v27, 3, v26\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_bitop3_b32 v42, v42, v46, v25 bitop3:0x96\n\tv_bitop3_b32 v43, v43, v47, v27 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_add_u32 v24, 0x428a2f98, v42\n\tv_add_u32 v26, 0x428a2f98, v43\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v42, v24, v24, 7\n\tv_alignbit_b32 v43, v26, v26, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_alignbit_b32 v46, v24, v24, 18\n\tv_alignbit_b32 v47, v26, v26, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_lshrrev_b32 v25, 3, v24\n\tv_lshrrev_b32 v27, 3, v26\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_bitop3_b32 v42, v42, v46, v25 bitop3:0x96\n\tv_bitop3_b32 v43, v43, v47, v27 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_add_u32 v24, 0x428a2f98, v42\n\tv_add_u32 v26, 0x428a2f98, v43\n\tv_alignbit_b32 v40, v20, v20, 7\n\tv_alignbit_b32 v41, v22, v22, 7\n\tv_alignbit_b32 v42, v24, v24, 7\n\tv_alignbit_b32 v43, v26, v26, 7\n\tv_alignbit_b32 v44, v20, v20, 18\n\tv_alignbit_b32 v45, v22, v22, 18\n\tv_alignbit_b32 v46, v24, v24, 18\n\tv_alignbit_b32 v47, v26, v26, 18\n\tv_lshrrev_b32 v21, 3, v20\n\tv_lshrrev_b32 v23, 3, v22\n\tv_lshrrev_b32 v25, 3, v24\n\tv_lshrrev_b32 v27, 3, v26\n\tv_bitop3_b32 v40, v40, v44, v21 bitop3:0x96\n\tv_bitop3_b32 v41, v41, v45, v23 bitop3:0x96\n\tv_bitop3_b32 v42, v42, v46, v25 bitop3:0x96\n\tv_bitop3_b32 v43, v43, v47, v27 bitop3:0x96\n\tv_add_u32 v20, 0x428a2f98, v40\n\tv_add_u32 v22, 0x428a2f98, v41\n\tv_add_u32 v24, 0x428a2f98, v42\n\tv_add_u32 v26, 0x428a2f98, v43" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
    }
    u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 r;
    asm volatile("v_xor_b32 %0, v20, v22\n\tv_xor_b32 %0, %0, v24\n\tv_xor_b32 %0, %0, v26" : "=v"(r));
    if (r == 0x12345678u) out[3] = r;
    if (t == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

typedef void (*kern_t)(u64 *, u32, int);
static void run(kern_t kern, const char *name, int nch, int per_step, int waves_per_simd, int cus, u64 *out)
{
    const int iters = 4000;
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 1u, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 7u, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    u64 h[2];
    (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    double instr = (double)iters * 8 * per_step * nch;
    printf("%-4s chains/lane=%d waves/SIMD=%d  one wave: %5.2f cycles between its instructions;  SIMD: %5.2f cycles per instruction (wall);  %6.2f cycles per chain step per SIMD  clk %.2f GHz\n",
           name, nch, waves_per_simd, (double)h[0] / instr, (ms * 1e-3) * ghz * 1e9 / (instr * waves_per_simd),
           (ms * 1e-3) * ghz * 1e9 / ((double)iters * 8 * nch * waves_per_simd), ghz);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("device %s  CUs %d\n", p.gcnArchName, cus);
    u64 *out;
    (void)hipMalloc(&out, 64);
    for (int w : {1, 2, 3, 4}) {
        run(k_mt1, "mt", 1, 3, w, cus, out);
        run(k_mt2, "mt", 2, 3, w, cus, out);
        run(k_mt4, "mt", 4, 3, w, cus, out);
        run(k_rot1, "rot", 1, 5, w, cus, out);
        run(k_rot2, "rot", 2, 5, w, cus, out);
        run(k_rot4, "rot", 4, 5, w, cus, out);
        printf("\n");
    }
    return 0;
}
