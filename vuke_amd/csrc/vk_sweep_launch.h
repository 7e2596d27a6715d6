// vk_sweep_launch.h -- host-side entry points of the sweep kernel instantiations (vk_sweep_tu.hip is compiled once per
// group of transform kinds, in parallel; each group launches the kinds it holds and reports the others as not handled).
#pragma once
#include "vk_kernels.cuh"

bool vk_launch_sweep_g0(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s); // milksad, mt64, raw keys
bool vk_launch_sweep_g1(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s); // sha256, double_sha256, md5, direct
bool vk_launch_sweep_g2(u32 kind, const vk_sweep_params &p, u32 grid, hipStream_t s); // lcg, xorshift
hipError_t vk_sweep_occupancy(int *blocks_per_cu);                                      // of the milksad instantiation
