// vk_sweep_launch.h -- host-side entry points of the sweep kernel instantiations (vk_sweep_tu.hip is compiled once per
// group of transform kinds, in parallel; each group launches the kinds it holds and reports the others as not handled).
// `waves`: the workgroups per CU the caller's grid is made for -- 3 selects the three-wave instantiation where one exists and the
// launch does not hash uncompressed keys.
#pragma once
#include "vk_kernels.cuh"

bool vk_launch_sweep_g0(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s, u32 waves); // milksad, mt64, raw keys
bool vk_launch_sweep_g1(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s, u32 waves); // sha256, double_sha256, md5, direct
bool vk_launch_sweep_g2(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s, u32 waves); // lcg, xorshift
u32 vk_sweep_design_waves(u32 kind, u32 flags); // workgroups per CU the instantiation for this launch is built for (2 or 3)
