// vk_sweep_tu.hip -- instantiations of vk_sweep_kernel<KIND, WAVES>, compiled once per group (-DVK_SWEEP_GROUP=0..2) so the three
// heavy translation units build side by side (build.sh).  sha256_chain and armory have no instantiation of their own: their
// keys are staged through HBM and swept as raw keys (vk_engine.hip: run_range_staged).
#include "../../include/vuke_hip.h"
#include "vk_sweep_launch.h"
#include <mutex>
#include <vector>

#ifndef VK_SWEEP_GROUP
#error "compile with -DVK_SWEEP_GROUP=0..2"
#endif

// Which instantiation runs a launch (vk_kernels.cuh: WAVES, UNCOMP, WIDE):
//   VK_FORM_TWO    <KIND, 2, true, false>   two waves per SIMD, both key forms behind the launch flags; grids of 256-lane workgroups, two
//                                           per CU (with the 53 KB schedule table of an uncompressed launch: 78 KB of LDS each)
//   VK_FORM_THREE  <KIND, 3, false, false>  three waves per SIMD, compressed keys only, none of the uncompressed code; three workgroups per CU
//   VK_FORM_WIDE   <KIND, 3, true, true>    three waves per SIMD for launches that hash uncompressed keys: ONE 768-lane workgroup per CU
//                                           around one schedule table (127 KB of LDS), tiles per wave
#if VK_SWEEP_GROUP == 0
u32 vk_sweep_design_form(u32 kind, u32 flags)
{
    if (flags & VK_F_UNCOMPRESSED) return VK_FORM_WIDE;
    switch (kind) {
    case VK_T_MILKSAD:
    case VK_T_MT64:
    case VK_T_RAWKEYS:
    case VK_T_LCG:
    case VK_T_XORSHIFT: return VK_FORM_THREE;
    default: return VK_FORM_TWO;
    }
}
#endif

// The 53 KB (two-wave form) / 127 KB (wide form) of LDS are beyond the 64 KB a kernel gets without asking: the attribute belongs to the
// device's copy of the function (engines of several GPUs in one process), is set once per device and function, and a failure is the
// launch's failure (it used to be discarded, and the flag array was racy and capped at 64 devices).
#ifndef VK_THREE_UNCOMP
#define VK_THREE_UNCOMP false /* A/B switch: true = the compressed-only three-wave form keeps the uncompressed code behind the launch flag (round 5's kernel) */
#endif
template <u32 KIND, int WAVES, bool UNCOMP, bool WIDE>
static hipError_t launch(const vk_sweep_params &p, u32 grid, hipStream_t s)
{
    // dynamic LDS: the schedule table of the uncompressed key's second SHA-256 block, only where such keys are hashed
    const size_t dyn = (UNCOMP && (p.flags & VK_F_UNCOMPRESSED)) ? (size_t)VK_TAIL_LDS_BYTES : 0;
    if (UNCOMP && dyn) {
        static std::mutex mu;
        static std::vector<char> attr_set; // per device ordinal
        int dev = 0;
        hipError_t rc = hipGetDevice(&dev);
        if (rc != hipSuccess) return rc;
        std::lock_guard<std::mutex> lk(mu);
        if ((size_t)dev >= attr_set.size()) attr_set.resize((size_t)dev + 1, 0);
        if (!attr_set[dev]) {
            rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&vk_sweep_kernel<KIND, WAVES, UNCOMP, WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)VK_TAIL_LDS_BYTES);
            if (rc != hipSuccess) return rc;
            attr_set[dev] = 1;
        }
    }
    hipLaunchKernelGGL((vk_sweep_kernel<KIND, WAVES, UNCOMP, WIDE>), dim3(grid), dim3(WIDE ? VK_WIDE_BLOCK : VK_BLOCK), dyn, s, p);
    return hipGetLastError();
}
// every kind exists in the two-wave and the wide form; VK_FORM_THREE only where it measured faster than two waves (design_form)
template <u32 KIND, bool HAS_THREE>
static hipError_t launch_form(const vk_sweep_params &p, u32 grid, hipStream_t s, u32 form)
{
    if (form == VK_FORM_WIDE && (p.flags & VK_F_UNCOMPRESSED)) return launch<KIND, 3, true, true>(p, grid, s);
    if constexpr (HAS_THREE) {
        if (form == VK_FORM_THREE && !(p.flags & VK_F_UNCOMPRESSED)) return launch<KIND, 3, VK_THREE_UNCOMP, false>(p, grid, s);
    }
    return launch<KIND, 2, true, false>(p, grid, s);
}

#if VK_SWEEP_GROUP == 0
bool vk_launch_sweep_g0(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s, u32 form, hipError_t *rc)
{
    switch (kind) {
    case VK_T_MILKSAD: *rc = launch_form<VK_T_MILKSAD, true>(p, grid, s, form); return true;
    case VK_T_MT64: *rc = launch_form<VK_T_MT64, true>(p, grid, s, form); return true;
    case VK_T_RAWKEYS: *rc = launch_form<VK_T_RAWKEYS, true>(p, grid, s, form); return true;
    default: return false;
    }
}
#elif VK_SWEEP_GROUP == 1
bool vk_launch_sweep_g1(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s, u32 form, hipError_t *rc)
{
    switch (kind) {
    case VK_T_SHA256: *rc = launch_form<VK_T_SHA256, false>(p, grid, s, form); return true;
    case VK_T_DOUBLE_SHA256: *rc = launch_form<VK_T_DOUBLE_SHA256, false>(p, grid, s, form); return true;
    case VK_T_MD5: *rc = launch_form<VK_T_MD5, false>(p, grid, s, form); return true;
    case VK_T_DIRECT: *rc = launch_form<VK_T_DIRECT, false>(p, grid, s, form); return true;
    default: return false;
    }
}
#elif VK_SWEEP_GROUP == 2
bool vk_launch_sweep_g2(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s, u32 form, hipError_t *rc)
{
    switch (kind) {
    case VK_T_LCG: *rc = launch_form<VK_T_LCG, true>(p, grid, s, form); return true;
    case VK_T_XORSHIFT: *rc = launch_form<VK_T_XORSHIFT, true>(p, grid, s, form); return true;
    default: return false;
    }
}
#endif
