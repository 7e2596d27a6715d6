// vk_sweep_launch.h -- host-side entry points of the sweep kernel instantiations (vk_sweep_tu.hip is compiled once per
// group of transform kinds, in parallel; each group launches the kinds it holds and reports the others as not handled).
// `form`: which instantiation the caller's grid and workspace are made for (VK_FORM_*); *rc = the launch's HIP status.
#pragma once
#include "vk_kernels.cuh"

#define VK_FORM_TWO 2u   /* 256-lane workgroups, two per CU, any key forms */
#define VK_FORM_THREE 3u /* 256-lane workgroups, three per CU, compressed keys only */
#define VK_FORM_WIDE 4u  /* 768-lane workgroups, one per CU, launches that hash uncompressed keys */

bool vk_launch_sweep_g0(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s, u32 form, hipError_t *rc); // milksad, mt64, raw keys
bool vk_launch_sweep_g1(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s, u32 form, hipError_t *rc); // sha256, double_sha256, md5, direct
bool vk_launch_sweep_g2(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s, u32 form, hipError_t *rc); // lcg, xorshift
u32 vk_sweep_design_form(u32 kind, u32 flags); // the form the automatic choice runs this launch in
