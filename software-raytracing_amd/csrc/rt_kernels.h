// rt_kernels.h - launch interface between the host side (rt_api.hip) and the kernels (rt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rt_mi355.h"

#include "rt_args.h"

namespace rt {
struct DevScene;

// jit: scene-specialised kernel for this scene (rt_jit.h) or nullptr
// items_grid_blocks: blocks of the persistent grid of the item form (A.items): what the device holds at once
void launch_path(int variant, const DevScene& S, const PathArgs& A, float4* out, int n_slices, unsigned long long* counters, hipStream_t st, hipFunction_t jit, int items_grid_blocks);
// optional BVH walk (rt_bvh.h, rt_bvh_device.h)
void launch_path_bvh(int variant, const DevScene& S, const PathArgs& A, float4* out, int n_slices, unsigned long long* counters, hipStream_t st);
void launch_add_image(float4* dst, const float4* src, int n_pixels, hipStream_t st);
void launch_sum_slices(const float4* part, int n_pixels, int n_slices, float4* out, hipStream_t st);
// jit: scene-specialised 02 kernel (rt_jit.h) or nullptr
void launch_whitted(bool soft, const DevScene& S, const WhittedArgs& A, float4* out, unsigned long long* counters, hipStream_t st, hipFunction_t jit);
// wavefront form of the Whitted kernels: tree walk -> one thread per shadow ray -> per-pixel combine (+ the monolithic kernel over overflowed pixels)
void launch_whitted_wavefront(bool soft, const DevScene& S, const WhittedArgs& A, float4* out, unsigned long long* counters, hipStream_t st, hipFunction_t jit, int grid_blocks, bool overflow_impossible);
void launch_resolve(const float4* accum, int width, int height, float divisor, int gamma, uint8_t* bgra, int stride, float* rgb, hipStream_t st);
} // namespace rt
