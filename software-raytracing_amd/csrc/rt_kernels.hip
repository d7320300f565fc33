// rt_kernels.hip - the tracing kernels (gfx950) and their launchers.  One thread per pixel(-sample slice); the device code
// lives in rt_path_device.h, rt_whitted_device.h and rt_bvh_device.h (also embedded as text for the scene-specialised
// hiprtc builds, rt_jit.hip).  No MFMA: the path is FP32 VALU-bound.
#include "rt_device.h"
#include "rt_strict.h"
#include "rt_kernels.h"
#include "rt_path_device.h"
#include "rt_whitted_device.h"
#include "rt_bvh_device.h"
#include <algorithm>

namespace rt {

// =================================================================================================
// K3/K4: path tracing (device code in rt_path_device.h).
// =================================================================================================
// REF: the reference's arithmetic for rays and spheres (RT_OPT_ARITHMETIC = RT_ARITH_REFERENCE; rt_device.h strict_sphere_roots)
template <int VARIANT, bool LDS, bool EARLY = false, bool REF = false>
__global__ __launch_bounds__(256) void k_path(DevScene SD, PathArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    path_body<VARIANT, LDS, EARLY, REF>(SD, A, out, counters);
}
template <int VARIANT, bool REF = false>
__global__ __launch_bounds__(256) void k_path_tiled(DevScene SD, PathArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    path_tiled_body<VARIANT, REF>(SD, A, out, counters);
}

// Queue form (rt_path_device.h): ray-queue compaction through LDS inside the workgroup.
template <int VARIANT, bool LDS, bool EARLY = false>
__global__ __launch_bounds__(256) void k_path_q(DevScene SD, PathArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    path_queue_body<VARIANT, LDS, EARLY>(SD, A, out, counters);
}

// Item form (rt_path_device.h): persistent waves pulling (pixel group x sample slice) units from a work counter.
template <int VARIANT, bool LDS, bool EARLY = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_path_items(DevScene SD, PathArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    path_items_body<VARIANT, LDS, EARLY>(SD, A, out, counters);
}

// Outcome-sorted form (rt_path_device.h): workgroups pull units, sort their paths by the query's outcome through LDS.
template <int VARIANT, bool LDS, bool EARLY = false>
__global__ __launch_bounds__(256) void k_path_sorted(DevScene SD, PathArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    path_sorted_body<VARIANT, LDS, EARLY>(SD, A, out, counters);
}

template <int VARIANT>
static void launch_path_v(const DevScene& S, const PathArgs& A, float4* out, int n_slices, unsigned long long* counters, hipStream_t st, hipFunction_t jit, int items_grid_blocks) {
    dim3 grid(((A.width + 15) / 16) * ((A.height + 15) / 16), n_slices);
    size_t bytes = (size_t)S.total_f4 * sizeof(float4);
    const bool queue = A.queue && bytes <= RT_LDS_SCENE_MAX_BYTES;      // queue form: LDS-resident scenes
    // item form: a persistent grid - as many blocks as the chip holds (the host's estimate; more would only find the counter exhausted), at most one wave per unit
    const bool items = A.items && !queue;
    if (items) grid = dim3((unsigned)std::max(1, std::min(items_grid_blocks, (A.n_units + 3) / 4)), 1);
    const bool sorted = A.sorted && !queue && !items;
    if (sorted) grid = dim3((unsigned)std::max(1, std::min(items_grid_blocks, A.n_units)), 1);
    if (jit && bytes <= RT_LDS_SCENE_MAX_BYTES) {   // scene-specialised k_path<VARIANT, true> / k_path_items / k_path_q (rt_jit.hip)
        DevScene s = S; PathArgs a = A;
        void* params[] = {&s, &a, &out, &counters};
        if (hipModuleLaunchKernel(jit, grid.x, grid.y, 1, 256, 1, 1, (unsigned)(bytes + (queue ? RT_Q_WS_BYTES : 0) + (items ? 4 * RT_ITEMS_WAVE_BYTES : 0) + (sorted ? RT_SORT_WS_BYTES : 0)), st, params, nullptr) == hipSuccess) return;
        (void)hipGetLastError();   // could not launch the specialised kernel: the generic one below renders the same image
    }
    if (sorted) {
        if (S.slot_bytes > 64 * 64) hipLaunchKernelGGL((k_path_sorted<VARIANT, true, true>), grid, dim3(256), bytes + RT_SORT_WS_BYTES, st, S, A, out, counters);
        else                        hipLaunchKernelGGL((k_path_sorted<VARIANT, true>), grid, dim3(256), bytes + RT_SORT_WS_BYTES, st, S, A, out, counters);
        return;
    }
    if (items) {
        const size_t ws = 4 * RT_ITEMS_WAVE_BYTES;
        if (A.debug_flags & 2)                hipLaunchKernelGGL((k_path_items<VARIANT, false>), grid, dim3(256), ws, st, S, A, out, counters);
        else if (S.slot_bytes > 64 * 64)      hipLaunchKernelGGL((k_path_items<VARIANT, true, true>), grid, dim3(256), bytes + ws, st, S, A, out, counters);
        else                                  hipLaunchKernelGGL((k_path_items<VARIANT, true>), grid, dim3(256), bytes + ws, st, S, A, out, counters);
        return;
    }
    if (queue && !(A.debug_flags & 2)) {
        if (S.slot_bytes > 64 * 64) hipLaunchKernelGGL((k_path_q<VARIANT, true, true>), grid, dim3(256), bytes + RT_Q_WS_BYTES, st, S, A, out, counters);
        else                        hipLaunchKernelGGL((k_path_q<VARIANT, true>), grid, dim3(256), bytes + RT_Q_WS_BYTES, st, S, A, out, counters);
        return;
    }
    if (A.ref_rays) {   // reference arithmetic: one lane per pixel; LDS-resident (with early-outs from 64 slots up) or tiled
        if (bytes > RT_LDS_SCENE_MAX_BYTES) hipLaunchKernelGGL((k_path_tiled<VARIANT, true>), grid, dim3(256), RT_TILE_BYTES + SLOT_BYTES, st, S, A, out, counters);
        else if (S.slot_bytes > 64 * 64)    hipLaunchKernelGGL((k_path<VARIANT, true, true, true>), grid, dim3(256), bytes, st, S, A, out, counters);
        else                                hipLaunchKernelGGL((k_path<VARIANT, true, false, true>), grid, dim3(256), bytes, st, S, A, out, counters);
        return;
    }
    if (A.debug_flags & 2)                   hipLaunchKernelGGL((k_path<VARIANT, false>), grid, dim3(256), 0, st, S, A, out, counters);   // A/B reference: scalar loads from the global blob
    else if (bytes <= RT_LDS_SCENE_MAX_BYTES && S.slot_bytes > 64 * 64)   // mid-size scene (beyond the specialised builds): LDS-resident, wave-level early-outs
        hipLaunchKernelGGL((k_path<VARIANT, true, true>), grid, dim3(256), bytes, st, S, A, out, counters);
    else if (bytes <= RT_LDS_SCENE_MAX_BYTES) hipLaunchKernelGGL((k_path<VARIANT, true>), grid, dim3(256), bytes, st, S, A, out, counters);  // small scene: LDS-resident
    else                                     hipLaunchKernelGGL((k_path_tiled<VARIANT>), grid, dim3(256), RT_TILE_BYTES + SLOT_BYTES, st, S, A, out, counters);   // large scene: LDS-tiled
}
void launch_path(int variant, const DevScene& S, const PathArgs& A, float4* out, int n_slices, unsigned long long* counters, hipStream_t st, hipFunction_t jit, int items_grid_blocks) {
    if (variant == 3) launch_path_v<3>(S, A, out, n_slices, counters, st, jit, items_grid_blocks);
    else              launch_path_v<4>(S, A, out, n_slices, counters, st, jit, items_grid_blocks);
}

// Optional BVH walk (rt_set_option(RT_OPT_ACCEL, RT_ACCEL_BVH)): same bounce loop, the scene query walks a tree.
template <int VARIANT, bool SIMPLE>
__global__ __launch_bounds__(256) void k_path_bvh(DevScene SD, PathArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    if (SIMPLE) path_bvh_body_simple<VARIANT>(SD, A, out, counters);
    else        path_bvh_body<VARIANT>(SD, A, out, counters);
}
void launch_path_bvh(int variant, const DevScene& S, const PathArgs& A, float4* out, int n_slices, unsigned long long* counters, hipStream_t st) {
    dim3 grid(((A.width + 15) / 16) * ((A.height + 15) / 16), n_slices);
    // the walk is bound by the latency of dependent fetches, so resident waves matter: the per-lane stack takes only what this tree needs
    const size_t lds = (size_t)(S.bvh_depth > 0 ? S.bvh_depth : 1) * 256 * sizeof(uint32_t);
    const bool simple = A.debug_flags & 32;     // one walk per lane as a function (A/B reference of the scheduled form)
    if (variant == 3) { if (simple) hipLaunchKernelGGL((k_path_bvh<3, true>), grid, dim3(256), lds, st, S, A, out, counters); else hipLaunchKernelGGL((k_path_bvh<3, false>), grid, dim3(256), lds, st, S, A, out, counters); }
    else              { if (simple) hipLaunchKernelGGL((k_path_bvh<4, true>), grid, dim3(256), lds, st, S, A, out, counters); else hipLaunchKernelGGL((k_path_bvh<4, false>), grid, dim3(256), lds, st, S, A, out, counters); }
}

// Sum the per-slice partial images in a fixed order (deterministic), out = float4(sum rgb, samples).
__global__ __launch_bounds__(256) void k_sum_slices(const float4* __restrict__ part, int n_pixels, int n_slices, float4* __restrict__ out) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pixels) return;
    float4 a = part[i];
    for (int s = 1; s < n_slices; ++s) { float4 b = part[(size_t)s * n_pixels + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    out[i] = a;
}
// dst += src (the root-gather form of the multi-device sum, rt_api.hip): fixed order, so deterministic
__global__ __launch_bounds__(256) void k_add_image(float4* __restrict__ dst, const float4* __restrict__ src, int n_pixels) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pixels) return;
    float4 a = dst[i], b = src[i];
    dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
void launch_add_image(float4* dst, const float4* src, int n_pixels, hipStream_t st) {
    hipLaunchKernelGGL(k_add_image, dim3((n_pixels + 255) / 256), dim3(256), 0, st, dst, src, n_pixels);
}
void launch_sum_slices(const float4* part, int n_pixels, int n_slices, float4* out, hipStream_t st) {
    hipLaunchKernelGGL(k_sum_slices, dim3((n_pixels + 255) / 256), dim3(256), 0, st, part, n_pixels, n_slices, out);
}

// =================================================================================================
// K1/K2: Whitted.  Render + TraceTay of 01 (Main.cpp:324-496) and 02 (Main.cpp:341-566).
// out = Mix(primary, Mix(refl, refr, wr), wp) is linear in the children, so the recursion tree is
// walked depth-first with a scalar weight per node; the one pending sibling per depth level lives
// in an LDS slot indexed by depth (bounded by max_depth), never in scratch memory.
// =================================================================================================
// 8 waves/SIMD (64 VGPRs, a few cold values spilled): this kernel is bound by dependent-issue latency, not registers:
// measured 02 4K/256 samples 10.5 ms at 4 waves, 9.3 at 5, 8.7 at 6, 8.3 at 8
__global__ __launch_bounds__(WBLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_whitted_soft(DevScene SD, WhittedArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    whitted_body<true>(SD, A, out, counters);
}
// 01 (one shadow ray per shaded hit and light): a frame is a fraction of a millisecond, bound by launch and latency, and the
// kernel takes the registers it needs (no scratch) rather than 8 waves per SIMD
__global__ __launch_bounds__(WBLOCK) void k_whitted_hard(DevScene SD, WhittedArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    whitted_body<false>(SD, A, out, counters);
}

// Wavefront form (rt_whitted_device.h): walk -> shadow rays, one thread each -> combine.
// the walk traces one closest-hit query per tree node and no shadow ray: it is latency-tolerant, so it takes the registers it needs (no scratch) instead of 8 waves per SIMD
__global__ __launch_bounds__(WBLOCK) void k_whitted_walk(DevScene SD, WhittedArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    whitted_body<false, true>(SD, A, out, counters);
}
template <bool SOFT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_whitted_shadow(DevScene SD, WhittedArgs A) { whitted_shadow_body<SOFT>(SD, A); }
template <bool SOFT>
__global__ __launch_bounds__(256) void k_whitted_combine(WhittedArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) { whitted_combine_body<SOFT>(A, out, counters); }

// A.wf_* set by the caller.  grid_blocks: what the device holds at once (stage 2 is a grid-stride loop over the requests).
void launch_whitted_wavefront(bool soft, const DevScene& S, const WhittedArgs& A, float4* out, unsigned long long* counters, hipStream_t st, hipFunction_t jit, int grid_blocks, bool overflow_impossible) {
    const int n_pixels = A.width * A.height;
    WhittedArgs B = A;
    B.stack_bytes = S.whitted_branches ? (A.max_depth + 2) * 8 * WBLOCK * (int)sizeof(float) : 0;
    B.shadow_lds = 0; B.only_marked = 0;
    hipLaunchKernelGGL(k_whitted_walk, dim3(((A.width + 15) / 16) * ((A.height + 7) / 8)), dim3(WBLOCK), (size_t)B.stack_bytes, st, S, B, out, counters);
    // threads of stage 2: 2^g per request - enough of them to occupy the chip (~2M) on small frames, but at least 4 samples each
    // (a thread's set-up - request, light, keys - is a chain of dependent loads: short threads are bound by its latency)
    int g = 0;
    if (soft) while (g < 6 && (8 << g) <= A.shadow_samples && (((unsigned long long)A.wf_base + A.wf_capacity) << g) < (2ULL << 20)) ++g;
    B.wf_group_log2 = g;
    const unsigned long long max_threads = ((unsigned long long)A.wf_base + A.wf_capacity) << g;
    const unsigned blocks = (unsigned)std::max<unsigned long long>(1ULL, std::min<unsigned long long>((max_threads + 255) / 256, (unsigned long long)grid_blocks * 4));
    WhittedArgs C = B;
    C.shadow_lds = soft && S.slot_bytes > 0 && S.slot_bytes <= RT_TILE_BYTES && S.srun_packed && !(A.debug_flags & 64);
    bool done = false;
    if (soft && jit) {   // scene-specialised stage 2: its any-hit query is generated code, nothing to stage
        WhittedArgs J = C; J.shadow_lds = 0;
        DevScene s = S;
        void* params[] = {&s, &J};
        done = hipModuleLaunchKernel(jit, blocks, 1, 1, 256, 1, 1, 0, st, params, nullptr) == hipSuccess;
        if (!done) (void)hipGetLastError();
    }
    if (!done) {
        const size_t lds = C.shadow_lds ? (size_t)S.slot_bytes + 256 : 0;
        if (soft) hipLaunchKernelGGL(k_whitted_shadow<true>, dim3(blocks), dim3(256), lds, st, S, C);
        else      hipLaunchKernelGGL(k_whitted_shadow<false>, dim3(blocks), dim3(256), 0, st, S, C);
    }
    if (soft) hipLaunchKernelGGL(k_whitted_combine<true>, dim3((n_pixels + 255) / 256), dim3(256), 0, st, B, out, counters);
    else      hipLaunchKernelGGL(k_whitted_combine<false>, dim3((n_pixels + 255) / 256), dim3(256), 0, st, B, out, counters);
    // pixels whose requests did not fit the queue: the monolithic kernel renders them whole (it exits at once everywhere else);
    // not needed when the queue holds the worst case (every pixel's walk shading max_depth + 1 nodes)
    if (!overflow_impossible) {
        WhittedArgs M = A; M.only_marked = 1;
        launch_whitted(soft, S, M, out, counters, st, nullptr);
    }
}

void launch_whitted(bool soft, const DevScene& S, const WhittedArgs& A, float4* out, unsigned long long* counters, hipStream_t st, hipFunction_t jit) {
    dim3 grid(((A.width + 15) / 16) * ((A.height + 7) / 8));
    WhittedArgs B = A;
    // the stack only ever holds the refraction child of a hit that ALSO reflects (0 < reflectToRefract < 1 on a material with
    // primaryToSecondary < 1): none of the reference's scenes has one, and then no LDS is spent on it (more blocks per CU)
    // per level: the record of the bottom-up colour evaluation (4 floats); branching trees: + the parked refraction child / the reflection's colour (8 floats)
    B.stack_bytes = (A.max_depth + 1) * (S.whitted_branches ? 12 : 4) * WBLOCK * (int)sizeof(float);
    B.shadow_lds = soft && S.slot_bytes > 0 && S.slot_bytes <= RT_TILE_BYTES && S.srun_packed && !(A.debug_flags & 64) &&
                   (size_t)B.stack_bytes + (size_t)S.slot_bytes + 256 <= 65536;      // depth 8 of a branching tree is 54 KB of records
    if (jit) {   // scene-specialised whitted_body<soft>: generated strict queries; 02: its shadow samples need no staged slots
        WhittedArgs J = B; J.shadow_lds = 0;
        DevScene s = S;
        void* params[] = {&s, &J, &out, &counters};
        if (hipModuleLaunchKernel(jit, grid.x, 1, 1, WBLOCK, 1, 1, (unsigned)J.stack_bytes, st, params, nullptr) == hipSuccess) return;
        (void)hipGetLastError();   // could not launch it: the generic kernel below renders the same image
    }
    size_t lds = (size_t)B.stack_bytes + (B.shadow_lds ? (size_t)S.slot_bytes + 256 : 0);   // + the leading empty slot (address 0 = "no hit") and a wide slot of slack for read-ahead
    if (soft) hipLaunchKernelGGL(k_whitted_soft, grid, dim3(WBLOCK), lds, st, S, B, out, counters);
    else      hipLaunchKernelGGL(k_whitted_hard, grid, dim3(WBLOCK), lds, st, S, B, out, counters);
}

// =================================================================================================
// K5: the tail of Render (03 Main.cpp:352-367; 01 Main.cpp:351-356 with gamma off, divisor 1):
// c = sum/divisor; gamma ? sqrt(c); Clamp(c,0,1)*255 truncated; bytes B,G,R,255.
// =================================================================================================
__global__ __launch_bounds__(256) void k_resolve(const float4* __restrict__ accum, int width, int height, float divisor, int gamma,
                                                 uint8_t* __restrict__ bgra, int stride, float* __restrict__ rgb) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= width * height) return;
    float4 a = accum[i];
    float c[3] = {a.x / divisor, a.y / divisor, a.z / divisor};
    if (rgb) { rgb[3 * (size_t)i] = c[0]; rgb[3 * (size_t)i + 1] = c[1]; rgb[3 * (size_t)i + 2] = c[2]; }
    if (bgra) {
        unsigned int q[3];
        for (int k = 0; k < 3; ++k) {
            float v = gamma ? sqrtf(c[k]) : c[k];
            v = (v < 0.0f) ? 0.0f : (1.0f < v) ? 1.0f : v;      // math::Clamp (Common/Math.hpp:1119-1122)
            v = v * 255.0f;
            q[k] = (v >= 0.0f) ? (unsigned int)v : 0U;          // static_cast<BYTE>: truncation
        }
        int yy = i / width, xx = i - yy * width;
        uchar4 pix = make_uchar4((unsigned char)q[2], (unsigned char)q[1], (unsigned char)q[0], 255);
        *reinterpret_cast<uchar4*>(bgra + (size_t)yy * stride + 4 * (size_t)xx) = pix;
    }
}
void launch_resolve(const float4* accum, int width, int height, float divisor, int gamma, uint8_t* bgra, int stride, float* rgb, hipStream_t st) {
    hipLaunchKernelGGL(k_resolve, dim3((width * height + 255) / 256), dim3(256), 0, st, accum, width, height, divisor, gamma, bgra, stride, rgb);
}

} // namespace rt
