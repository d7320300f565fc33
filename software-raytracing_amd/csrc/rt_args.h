// rt_args.h - kernel argument records shared by the host side, the kernels and the scene-specialised (hiprtc) builds.
// Device-compilable without any include (hiprtc has no headers).
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#define RT_LDS_SCENE_MAX_BYTES (48 * 1024)
#define RT_BVH_STACK 48   /* optional BVH walk: per-lane stack entries in LDS; the builder refuses deeper trees */

// LDS workspace of the outcome-sorted form (path_sorted_body, rt_path_device.h) behind the staged scene blob
#define RT_SORT_MAX_LOG2 7                                                  /* up to 128 pixels per workgroup unit */
#define RT_SORT_XCH_BYTES (256 * 48)
#define RT_SORT_WS_BYTES (2 * RT_SORT_XCH_BYTES + (1 << RT_SORT_MAX_LOG2) * (24 + 8) + 64)      /* 2 exchange buffers, sums, pixel table, counters */

namespace rt {
// Ray counters: one 64-bit atomic per wave and counter; 130k waves hitting ONE address cost ~3 ms at 4K (the whole 01
// frame takes less), so the atomics are spread over banks (bank = block index mod RT_COUNTER_BANKS), summed on the host.
#define RT_COUNTER_BANKS 256
#define RT_COUNTER_WORDS (2 * RT_COUNTER_BANKS)
// behind the ray counters, in the same allocation (cleared with them before every launch): the work counter of the item
// kernels (path_items_body, rt_path_device.h) - the index of the next unit of work a wave may take
#define RT_WORK_COUNTER RT_COUNTER_WORDS
#define RT_COUNTER_ALLOC_WORDS (RT_COUNTER_WORDS + 8)

struct PathArgs {
    int width, height;
    int sample_begin, sample_end;   // samples [begin,end) of every pixel
    int slice_len;                  // samples per blockIdx.y slice
    int max_depth;
    int jitter;                     // spp > 1 (03 Main.cpp:331)
    int regen_threshold;            // idle lanes per wave that trigger a regeneration pass
    int debug_flags;                // bit 1 (2): no LDS staging (A/B switch); bit 2 (4): accumulator w = rays of the pixel
    int queue;                      // LDS-resident scenes: the queue form of the bounce loop (path_queue_body) instead of per-lane regeneration
    // item form (path_items_body): a unit of work = 2^unit_log2 consecutive pixels (row-major) x one sample slice; n_groups
    // pixel groups per slice, n_units = n_groups x slices units in all, handed out through counters[RT_WORK_COUNTER]
    int items, unit_log2, n_groups, n_units;
    int sorted;                     // the outcome-sorted form (path_sorted_body): units belong to workgroups
    int unit_order;                 // item form: order in which the pixel groups are handed out: 0 row-major, 1 bottom-up (last group first)
    int ref_rays;                   // RT_ARITH_REFERENCE: host-side choice of the kernel instantiation (the device code takes it as a template parameter)
    int strict_point;               // the scene has slots in the reference's operation sequence (coplanar elements): hit points are o + d*t in two roundings, as the reference computes them
    int norm_dirs;                  // the scene has a giant sphere (radius > RT_GIANT_SPHERE_RATIO x tMin): scattered directions are normalised in the reference's operation order
    uint32_t seed;
    float t_min, t_min_pos, t_max;  // t_min_pos = max(t_min, smallest positive float): "t > 0 && t >= tMin" in one compare
    float inv_w2, inv_h2, kx, ky;   // 2/w, 2/h, tan(fov/2)*w/h, tan(fov/2)
    float r[9];                     // GetRotationMat(orientation), row-major
    float px, py, pz;
};

struct WhittedArgs {
    int width, height, max_depth, shadow_samples;
    uint32_t seed;
    float w, h, tan_half;           // float(width), float(height), tanf(fovRadians/2) (01 Main.cpp:329-341)
    int stack_bytes, shadow_lds;    // set by launch_whitted: LDS bytes of the recursion stack; slot layout staged behind it for 02's shadow samples
    int debug_flags;                // bit 6 (64): 02's shadow samples on the global-memory layout even when the slots would fit LDS (A/B switch)
    // wavefront form (rt_whitted_device.h): the tree walk emits one REQUEST per (shaded hit, light) instead of tracing the shadow
    // rays itself; requests: 3 float4 each, wf_prev / wf_occ one word each, wf_head one word per pixel, wf_counter one word
    float4* wf_req; uint32_t* wf_prev; uint32_t* wf_occ; uint32_t* wf_head; uint32_t* wf_counter;
    uint32_t wf_base, wf_capacity;  // slots [0, wf_base): one per (pixel, light) for primary hits; [wf_base, wf_base + wf_capacity): deeper nodes, allocated through wf_counter
    int wf_group_log2;              // stage 2, 02: 2^wf_group_log2 threads share a request's samples
    int only_marked;                // monolithic kernel: render only the pixels whose wf_head says "overflowed"
};
#define RT_WF_NONE 0xFFFFFFFFU       /* wf_head / wf_prev: end of a pixel's request chain */
#define RT_WF_OVERFLOW 0xFFFFFFFEU   /* wf_head: the request buffer was full; the monolithic kernel renders this pixel */
} // namespace rt
