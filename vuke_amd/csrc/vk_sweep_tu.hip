// vk_sweep_tu.hip -- instantiations of vk_sweep_kernel<KIND>, compiled once per group (-DVK_SWEEP_GROUP=0..2) so the three
// heavy translation units build side by side (build.sh).  sha256_chain and armory have no instantiation of their own: their
// keys are staged through HBM and swept as raw keys (vk_engine.hip: run_range_staged).
#include "../../include/vuke_hip.h"
#include "vk_sweep_launch.h"

#ifndef VK_SWEEP_GROUP
#error "compile with -DVK_SWEEP_GROUP=0..2"
#endif

template <u32 KIND>
static void launch(const vk_sweep_params &p, u32 grid, hipStream_t s)
{
    // dynamic LDS: the schedule table of the uncompressed key's second SHA-256 block, only where such keys are hashed (two resident
    // workgroups then hold 2 x 78 KB of the CU's 160 KB; the register budget allows two either way)
    const size_t dyn = (p.flags & VK_F_UNCOMPRESSED) ? (size_t)VK_TAIL_LDS_BYTES : 0;
    static bool attr_set[64] = {}; // per device: the attribute belongs to the device's copy of the function (engines of several GPUs in one process)
    int dev = 0;
    if (dyn && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&vk_sweep_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)VK_TAIL_LDS_BYTES);
        (void)hipGetLastError();
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(vk_sweep_kernel<KIND>, dim3(grid), dim3(VK_BLOCK), dyn, s, p);
}

#if VK_SWEEP_GROUP == 0
bool vk_launch_sweep_g0(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s)
{
    switch (kind) {
    case VK_T_MILKSAD: launch<VK_T_MILKSAD>(p, grid, s); return true;
    case VK_T_MT64: launch<VK_T_MT64>(p, grid, s); return true;
    case VK_T_RAWKEYS: launch<VK_T_RAWKEYS>(p, grid, s); return true;
    default: return false;
    }
}
hipError_t vk_sweep_occupancy(int *blocks_per_cu)
{
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, vk_sweep_kernel<VK_T_MILKSAD>, VK_BLOCK, 0);
}
#elif VK_SWEEP_GROUP == 1
bool vk_launch_sweep_g1(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s)
{
    switch (kind) {
    case VK_T_SHA256: launch<VK_T_SHA256>(p, grid, s); return true;
    case VK_T_DOUBLE_SHA256: launch<VK_T_DOUBLE_SHA256>(p, grid, s); return true;
    case VK_T_MD5: launch<VK_T_MD5>(p, grid, s); return true;
    case VK_T_DIRECT: launch<VK_T_DIRECT>(p, grid, s); return true;
    default: return false;
    }
}
#elif VK_SWEEP_GROUP == 2
bool vk_launch_sweep_g2(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s)
{
    switch (kind) {
    case VK_T_LCG: launch<VK_T_LCG>(p, grid, s); return true;
    case VK_T_XORSHIFT: launch<VK_T_XORSHIFT>(p, grid, s); return true;
    default: return false;
    }
}
#endif
