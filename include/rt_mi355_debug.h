/* rt_mi355_debug.h - diagnostic exports of librt_mi355.so: test hooks, A/B switches and host-logic probes.
 *
 * NOT part of the drop-in boundary (include/rt_mi355.h): nothing here replaces a reference interface, a host that renders
 * never needs it, and the switches may change between versions.  The tests, bench.py and tools/ steer and inspect the
 * kernels through these; they are declared here so that every exported symbol of the library has a declaration. */
#ifndef RT_MI355_DEBUG_H
#define RT_MI355_DEBUG_H
#include "rt_mi355.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Debug flags of a context (bit set, 0 = defaults):
 *      2  read the scene from global memory with scalar loads instead of its LDS copy (A/B reference of the LDS kernels)
 *      4  the accumulator's w channel holds the pixel's scene-query count instead of its sample count (parity diagnostics)
 *      8  never use a scene-specialised (hiprtc) kernel          16  specialise at the FIRST render of a scene
 *     32  BVH mode: the walk as a per-lane function (A/B reference of the scheduled walk)
 *     64  02: shadow samples on the global-memory layout even when the slots fit LDS
 *    256  queue form of the bounce loop (path_queue_body)       512  one lane per pixel, per-lane regeneration (path_body)
 *   1024  item / sorted form whatever the frame size (default: only frames that can occupy the chip)
 *   2048  outcome-sorted form of the bounce loop (path_sorted_body)
 *   4096  02: the monolithic kernel even without a scene-specialised build
 *   8192  02: the wavefront form of the Whitted kernel (tree walk -> shadow rays, several threads per request -> per-pixel combine) */
int32_t rt_debug_set_flags(rt_ctx* ctx, int32_t flags);
/* idle lanes per wave that trigger a regeneration pass; < 0 restores the default */
int32_t rt_debug_set_regen_threshold(rt_ctx* ctx, int32_t n);
/* which kernel rendered the last path image: 3 BVH walk, 2 scene-specialised, 1 generic LDS-resident, 0 tiled / global;
 * note: when not specialised, why */
int32_t rt_debug_last_path_kernel(const rt_ctx* ctx, char* note, int32_t cap);
/* which form of the bounce loop: 0 one lane per pixel, 1 queue, 2 item form, 3 outcome-sorted; unit_log2: log2(pixels per unit) */
int32_t rt_debug_last_path_form(const rt_ctx* ctx, int32_t* unit_log2);
/* 2 scene-specialised (02 only), 1 generic */
int32_t rt_debug_last_whitted_kernel(const rt_ctx* ctx);
/* 1 wavefront form (tree walk -> one thread per shadow ray -> per-pixel combine), 0 monolithic kernel */
int32_t rt_debug_last_whitted_form(const rt_ctx* ctx);
/* how a device group sums its accumulators: 1 RCCL all-reduce, 0 root-gather form / single device; note: why */
int32_t rt_debug_group_mode(const rt_ctx* ctx, char* note, int32_t cap);

/* ---- host logic, no GPU needed ---- */
/* Generates and compiles the scene-specialised kernel source (loading the module needs a GPU: returns 1 when loaded, 0
 * otherwise with the reason in note).  variant: 3 / 4 path kernels, 2 the 02 kernel; + 8 queue form, + 16 item form, + 32 sorted form. */
int32_t rt_debug_jit_probe(const rt_scene* scene, int32_t variant, char* note, int32_t cap);
/* Builds the BVH of a scene and checks it; out[0..4] = {nodes, leaf entries, unbounded slots, depth, slots}; returns violations. */
int32_t rt_debug_bvh_probe(const rt_scene* scene, int32_t variant, int32_t* out);
/* Slots that take the reference's own operation sequence (coplanar with an element of other arithmetic);
 * out[0..2] = {planes, rectangles / box faces, triangles}; returns their total or an error code < 0. */
int32_t rt_debug_strict_slots(const rt_scene* scene, int32_t variant, int32_t* out);
/* share i of n of the sample range [s0, s1) as rt_create_multi cuts it */
int32_t rt_debug_group_share(int32_t s0, int32_t s1, int32_t i, int32_t n, int32_t* begin, int32_t* end);

/* Self-test of the run-time RCCL binding on one device: 0 ok, 1 library / symbols missing, 2 an RCCL call failed, 3 HIP error,
 * 4 wrong result; msg: detail. */
int32_t rt_debug_rccl_selftest(int32_t device, int32_t n, char* msg, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355_DEBUG_H */
