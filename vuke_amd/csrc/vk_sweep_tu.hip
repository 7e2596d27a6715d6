// vk_sweep_tu.hip -- instantiations of vk_sweep_kernel<KIND, WAVES>, compiled once per group (-DVK_SWEEP_GROUP=0..2) so the three
// heavy translation units build side by side (build.sh).  sha256_chain and armory have no instantiation of their own: their
// keys are staged through HBM and swept as raw keys (vk_engine.hip: run_range_staged).
#include "../../include/vuke_hip.h"
#include "vk_sweep_launch.h"

#ifndef VK_SWEEP_GROUP
#error "compile with -DVK_SWEEP_GROUP=0..2"
#endif

// The waves per SIMD (= workgroups per CU) the instantiation that will run this launch is built for (vk_kernels.cuh: WAVES).
// A launch that hashes uncompressed keys asks for the 53 KB schedule table of their second SHA-256 block: 78 KB of LDS per workgroup,
// two workgroups per CU whatever the registers allow -- such launches, and the sha256 family (whose BASELINE config hashes both key
// forms), keep the two-wave instantiation.
#if VK_SWEEP_GROUP == 0
u32 vk_sweep_design_waves(u32 kind, u32 flags)
{
    if (flags & VK_F_UNCOMPRESSED) return 2;
    switch (kind) {
    case VK_T_MILKSAD:
    case VK_T_MT64:
    case VK_T_RAWKEYS:
    case VK_T_LCG:
    case VK_T_XORSHIFT: return 3;
    default: return 2;
    }
}
#endif

template <u32 KIND, int WAVES>
static void launch(const vk_sweep_params &p, u32 grid, hipStream_t s)
{
    // dynamic LDS: the schedule table of the uncompressed key's second SHA-256 block, only where such keys are hashed (two resident
    // workgroups then hold 2 x 78 KB of the CU's 160 KB)
    const size_t dyn = (p.flags & VK_F_UNCOMPRESSED) ? (size_t)VK_TAIL_LDS_BYTES : 0;
    static bool attr_set[64] = {}; // per device: the attribute belongs to the device's copy of the function (engines of several GPUs in one process)
    int dev = 0;
    if (dyn && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&vk_sweep_kernel<KIND, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)VK_TAIL_LDS_BYTES);
        (void)hipGetLastError();
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((vk_sweep_kernel<KIND, WAVES>), dim3(grid), dim3(VK_BLOCK), dyn, s, p);
}
// the three-wave instantiation where the caller's grid is made for it, the two-wave one otherwise
template <u32 KIND>
static void launch23(const vk_sweep_params &p, u32 grid, hipStream_t s, u32 waves)
{
    if (waves >= 3 && !(p.flags & VK_F_UNCOMPRESSED)) launch<KIND, 3>(p, grid, s);
    else launch<KIND, 2>(p, grid, s);
}

#if VK_SWEEP_GROUP == 0
bool vk_launch_sweep_g0(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s, u32 waves)
{
    switch (kind) {
    case VK_T_MILKSAD: launch23<VK_T_MILKSAD>(p, grid, s, waves); return true;
    case VK_T_MT64: launch23<VK_T_MT64>(p, grid, s, waves); return true;
    case VK_T_RAWKEYS: launch23<VK_T_RAWKEYS>(p, grid, s, waves); return true;
    default: return false;
    }
}
#elif VK_SWEEP_GROUP == 1
bool vk_launch_sweep_g1(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s, u32 waves)
{
    (void)waves;
    switch (kind) {
    case VK_T_SHA256: launch<VK_T_SHA256, 2>(p, grid, s); return true;
    case VK_T_DOUBLE_SHA256: launch<VK_T_DOUBLE_SHA256, 2>(p, grid, s); return true;
    case VK_T_MD5: launch<VK_T_MD5, 2>(p, grid, s); return true;
    case VK_T_DIRECT: launch<VK_T_DIRECT, 2>(p, grid, s); return true;
    default: return false;
    }
}
#elif VK_SWEEP_GROUP == 2
bool vk_launch_sweep_g2(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s, u32 waves)
{
    switch (kind) {
    case VK_T_LCG: launch23<VK_T_LCG>(p, grid, s, waves); return true;
    case VK_T_XORSHIFT: launch23<VK_T_XORSHIFT>(p, grid, s, waves); return true;
    default: return false;
    }
}
#endif
