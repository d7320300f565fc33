/*
 * rt_mi355.h - C ABI of librt_mi355.so, the MI355X (gfx950) replacement for the per-pixel
 * tracing loop of darkoffalex/software-raytracing.
 *
 * The reference has no FFI: its seam is the free function Render(...) of each example
 * (Sources/01_Basic/Main.cpp:324, Sources/02_SoftShadows/Main.cpp:341,
 * Sources/03_PathTracingBasics/Main.cpp:310, Sources/04_PathTracingLights/Main.cpp:324) and,
 * below it, TraceTay(...) (01: Main.cpp:372, 02: :389, 03: :380, 04: :431).  The host-side C++
 * mirror of those functions (software-raytracing_amd/host/) flattens its scene objects into
 * the POD records declared here and calls these entry points; INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Plain pointers and sizes only; no exceptions cross this boundary; every function that can
 * fail returns an int (RT_OK or a negative RT_ERR_*), with text available from
 * rt_last_error().  The caller owns every host buffer it passes; the library owns the
 * device memory inside rt_ctx and keeps no caller pointer after a call returns.
 */
#ifndef RT_MI355_H
#define RT_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 2   /* 2: rt_stats.allreduce_ms, rt_create_multi */

/* ---- return codes -------------------------------------------------------------------- */
#define RT_OK               0
#define RT_ERR_BAD_ARG     -1   /* null pointer, non-positive size, index out of range      */
#define RT_ERR_UNSUPPORTED -2   /* primitive/material type the kernels do not know         */
#define RT_ERR_HIP         -3   /* a HIP runtime call failed (text in rt_last_error)       */
#define RT_ERR_NO_DEVICE   -4   /* no usable gfx950 device / extension not built for it    */
#define RT_ERR_TOO_LARGE   -5   /* size beyond a documented maximum                        */

/* ---- scene records ------------------------------------------------------------------- */

/* Primitive kinds.  Sphere/Plane: 01/02 Sphere.hpp, Plane.hpp; 03 Sphere.hpp; 04 Scene/Sphere.hpp,
 * Scene/Plane.hpp.  Rectangle: 04 Scene/Rectangle.hpp.  Box: 04 Scene/Box.hpp.  Triangle: semantics of
 * Ray::intersectsTriangle (Common/Ray.hpp:150-212), which no reference scene element wraps. */
enum rt_prim_type {
    RT_PRIM_SPHERE   = 0,  /* f[0..2] centre, f[3] radius                                    */
    RT_PRIM_PLANE    = 1,  /* f[0..2] point on plane, f[3..5] unit normal (the reference's
                              Plane constructors normalise; this ABI takes it as given)     */
    RT_PRIM_RECT     = 2,  /* f[0..2] centre, f[3..4] width,height, f[5..7] Euler orientation
                              in degrees (04 Scene/Rectangle.hpp:31-36)                      */
    RT_PRIM_TRIANGLE = 3,  /* f[0..2] v0, f[3..5] v1, f[6..8] v2                             */
    RT_PRIM_BOX      = 4   /* f[0..2] centre, f[3..5] sizes (x,y,z), f[6..8] Euler orientation in
                              degrees; 04 Scene/Box.hpp:34-40 with its six-face test :55-158,
                              including the y range [-sy/2, +sz/2] of the two Z faces (:88,:95);
                              path renders only                                              */
};

#define RT_PRIM_FLAG_FLIP_NORMALS 1u /* 03 Sphere 'inverted' (Sphere.hpp:65-67), 04 Sphere/Box 'flipNormals' */

typedef struct rt_prim {
    int32_t  type;        /* enum rt_prim_type                      */
    int32_t  material;    /* index into rt_scene.materials, or -1 = no material (black)    */
    uint32_t flags;       /* RT_PRIM_FLAG_*                          */
    int32_t  reserved;
    float    f[12];
} rt_prim;                /* 64 bytes */

/* Material kinds.  PHONG is the 01/02 aggregate (01_Basic/Types.h:9-23); the others are the
 * 03 classes (Materials.hpp) and the 04 classes (Materials/Diffuse|Metal|Refractive|Light.hpp);
 * which sampler/attenuation rule a DIFFUSE or METAL uses is decided by rt_path_params.variant. */
enum rt_material_type {
    RT_MAT_PHONG      = 0,
    RT_MAT_DIFFUSE    = 1,
    RT_MAT_METAL      = 2,
    RT_MAT_DIELECTRIC = 3,  /* 03 MaterialDielectric == 04 materials::Refractive             */
    RT_MAT_LIGHT      = 4   /* 04 materials::Light: emits 'albedo' from the front face only  */
};

typedef struct rt_material {
    int32_t type;                 /* enum rt_material_type                                  */
    float   albedo[3];            /* PHONG/DIFFUSE/METAL albedo; LIGHT emitted colour        */
    float   specular_intensity;   /* PHONG                                                   */
    float   shininess;            /* PHONG                                                   */
    float   primary_to_secondary; /* PHONG                                                   */
    float   reflect_to_refract;   /* PHONG                                                   */
    float   refraction_eta;       /* PHONG                                                   */
    float   roughness;            /* METAL                                                   */
    float   refraction_index;     /* DIELECTRIC                                              */
    float   reserved;
} rt_material;                    /* 48 bytes */

/* 01/02 LightSource (01_Basic/Types.h:28-36). */
typedef struct rt_light {
    float position[3];
    float color[3];
    float radius;                 /* used by the 02 soft-shadow sampler only                 */
    float reserved;
} rt_light;                       /* 32 bytes */

typedef struct rt_scene {
    const rt_prim*     prims;      int32_t n_prims;
    const rt_material* materials;  int32_t n_materials;
    const rt_light*    lights;     int32_t n_lights;     /* Whitted renders only             */
} rt_scene;

/* The reference has no camera type: 01/02 Render take fov only (origin, looking down -Z);
 * 03/04 Render take fov, position and Euler orientation in degrees. */
typedef struct rt_camera {
    float fov_deg;
    float position[3];
    float orientation_deg[3];
} rt_camera;

/* Output of the host-buffer entry points.  bgra: 8-bit B,G,R,255 per pixel, row-major, top-left
 * origin, exactly what the reference's Render writes into ImageBuffer<RGBQUAD>
 * (01 Main.cpp:351-356, 04 Main.cpp:390-395).  rgb (optional): linear float R,G,B per pixel
 * before gamma/quantisation (mean over samples for path renders). Either may be NULL. */
typedef struct rt_image {
    uint8_t* bgra;
    float*   rgb;
    int32_t  width;
    int32_t  height;
    int32_t  bgra_stride_bytes;   /* 0 = width*4                                             */
} rt_image;

/* ---- render parameters --------------------------------------------------------------- */

#define RT_WHITTED_MAX_DEPTH_LIMIT 8
#define RT_MAX_SHADOW_SAMPLES      1024

typedef struct rt_whitted_params {
    int32_t  max_depth;       /* MAX_RECURSION_DEPTH: 4 in 01/02 (Main.cpp:12 / :14)          */
    int32_t  shadow_samples;  /* 0 = example 01 (one hard shadow ray per light);
                                 >=1 = example 02 (that many disk-light samples, 16 there)   */
    uint32_t seed;            /* counter-based RNG seed (02 only)                            */
    uint32_t reserved;
} rt_whitted_params;

enum rt_path_variant {
    RT_PATH_SKY    = 3,  /* 03_PathTracingBasics: sky gradient on miss, uniform-in-angle
                            hemisphere sampler, attenuation = albedo                         */
    RT_PATH_LIGHTS = 4   /* 04_PathTracingLights: black on miss, uniform-solid-angle sampler,
                            attenuation = albedo*cos, emissive materials                     */
};

typedef struct rt_path_params {
    int32_t  variant;       /* enum rt_path_variant                                          */
    int32_t  spp;           /* samples per pixel of the whole image (divisor; >1 => jitter)  */
    int32_t  max_depth;     /* MAX_RECURSION_DEPTH: 24 in 03, 6 in 04                         */
    uint32_t seed;
    int32_t  sample_begin;  /* this call renders samples [sample_begin, sample_end) ...      */
    int32_t  sample_end;    /* ... of every pixel; 0,0 means [0, spp)                         */
    float    t_min;         /* 0.001 in 03 (Main.cpp:392), 0.01 in 04 (Main.cpp:447)          */
    float    t_max;         /* 1000 in both                                                  */
} rt_path_params;

typedef struct rt_stats {
    uint64_t rays;          /* closest-hit scene queries of the last render                  */
    uint64_t shadow_rays;   /* any-hit (shadow) scene queries of the last render             */
    uint64_t samples;       /* pixel-samples of the last render                              */
    double   kernel_ms;     /* HIP-event time of the tracing kernel(s) of the last render    */
    double   total_ms;      /* wall time of the last call incl. uploads and read-back        */
    double   allreduce_ms;  /* device groups: GPU time of the sum of the per-GPU accumulators (else 0); for a
                               group, rays/samples add up over its devices and kernel_ms is the slowest device's */
} rt_stats;

/* ---- context ------------------------------------------------------------------------- */
typedef struct rt_ctx rt_ctx;

/* Creates a context bound to one HIP device (device_id as HIP numbers it; -1 = current).
 * Returns NULL on failure (no GPU, wrong architecture); rt_last_error(NULL) has the reason.
 * Replaces nothing in the reference (it has no device); it is what the four Render wrappers
 * call once per process. */
rt_ctx* rt_create(int32_t device_id);
/* A context over several GPUs of one node (one process drives them all).  Replaces the reference's 8-thread split of 04
 * Render (04 Main.cpp:399-420): rt_render_path on such a context cuts the frame's samples into n_devices contiguous ranges,
 * device_ids[i] renders range i of EVERY pixel into its own accumulator (all devices side by side, one stream each), the
 * accumulators are combined with ONE RCCL all-reduce(sum, float32) over xGMI (librccl is loaded at this call, with
 * dlopen), and device_ids[0] resolves the frame.  The result does not depend on n_devices except for the order of the
 * float additions.  A device id may be listed more than once (rehearsal of the group code on one GPU), RCCL may be
 * missing or fail to initialise, RT_MI355_NO_RCCL may be set: the sums then go through copies to device_ids[0], added in a
 * fixed order (root-gather form) - never an error.  Whitted renders and the device-buffer entry points use device_ids[0]
 * only.  Returns NULL on failure (null list, count outside [1, 64], a device that rt_create refuses; rt_last_error(NULL));
 * rt_destroy frees the whole group.
 * STATE OF VERIFICATION: the fan-out, the root-gather sum and the resolve are tested on hardware (a group that lists one GPU
 * twice); RCCL's run-time binding is tested with a one-rank communicator.  The all-reduce across DISTINCT devices has never
 * run on hardware (no multi-GPU node was available to any round): parity of that path is unpinned. */
rt_ctx* rt_create_multi(const int32_t* device_ids, int32_t n_devices);
void    rt_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx);   /* ctx may be NULL for rt_create failures    */
int32_t rt_get_stats(const rt_ctx* ctx, rt_stats* out);
int32_t rt_abi_version(void);

/* Options of a context.  No counterpart in the reference.
 * RT_OPT_ACCEL: how the path renders (rt_render_path, rt_path_accumulate_device) query the scene.
 *   RT_ACCEL_NONE (default): brute force over every primitive in scene order, the reference's algorithm
 *                            (03 Utils.h:255-271, 04 Utils.h:362-390) and the one the roofline figures are quoted on.
 *   RT_ACCEL_BVH:            a bounding-volume hierarchy over spheres, rectangles, box faces and triangles (planes are
 *                            tested by every ray).  Same image, bit for bit - exact ties (last element wins) and the
 *                            sphere tMax rule (Common/Ray.hpp:127-134) included - in far fewer primitive tests on
 *                            large scenes.  Whitted renders ignore it. */
#define RT_OPT_ACCEL  1
#define RT_ACCEL_NONE 0
#define RT_ACCEL_BVH  1
/* RT_OPT_ARITHMETIC: how closely the path renders follow the reference's floating-point operation sequence.
 *   RT_ARITH_FAST (default):  fused multiply-adds, hardware reciprocals and roots, directions that are unit by construction.  The
 *                             same paths as the reference's arithmetic in all but the cases the last bits decide; planar elements
 *                             that coincide and spheres of radius > 30000 tMin (03's ground sphere) are tested in the
 *                             reference's own operation sequence, because there those bits decide OFTEN.
 *   RT_ARITH_REFERENCE:       every ray direction normalised as Ray's constructor does (Common/Ray.hpp:33-34), every sphere
 *                             tested with Ray::intersectsSphere's own operations (Common/Ray.hpp:105-121: a = d.d, true
 *                             divisions, correctly rounded root), sphere normals and hit points rounded as the reference rounds them.  For scenes
 *                             of small spheres far from where the rays start (|o - c| >> r) at a small tMin, where a ray
 *                             re-hits the surface it leaves at a rate the rounding sets: scene queries per sample then follow
 *                             the reference to ~3e-4 instead of ~1e-2.  Costs about 40 % on 03; brute force and one lane per
 *                             pixel only (RT_ACCEL_BVH is refused).  Whitted renders are reference arithmetic always. */
#define RT_OPT_ARITHMETIC  2
#define RT_ARITH_FAST      0
#define RT_ARITH_REFERENCE 1
int32_t rt_set_option(rt_ctx* ctx, int32_t option, int32_t value);

/* Fills params with the committed constants of the named example (1..4). */
int32_t rt_whitted_params_default(int32_t example, rt_whitted_params* out);
int32_t rt_path_params_default(int32_t example, rt_path_params* out);

/* ---- host-buffer entry points (the drop-in seam) ------------------------------------- */

/* Replaces the pixel loop + TraceTay recursion of 01 Render (Main.cpp:324-359 -> :372-496)
 * and 02 Render (Main.cpp:341-376 -> :389-566).  camera->position/orientation are ignored
 * by the reference for these examples and must be zero. Blocking. */
int32_t rt_render_whitted(rt_ctx* ctx, const rt_scene* scene, const rt_camera* camera,
                          const rt_whitted_params* params, const rt_image* out);

/* Replaces 03 Render (Main.cpp:310-370 -> :380-435) and 04 Render incl. its 8-thread split
 * (Main.cpp:324-421 -> :431-502). Blocking. */
int32_t rt_render_path(rt_ctx* ctx, const rt_scene* scene, const rt_camera* camera,
                       const rt_path_params* params, const rt_image* out);

/* ---- progressive frame (interactive preview) ----------------------------------------- */
/* The reference's main() renders the whole frame, then shows it once (Render -> PresentFrame, 01 Main.cpp:206-215 and
 * :272-310; the same in 02-04).  For an interactive viewer the path-traced frame is built up in passes instead: the running
 * per-pixel sums stay in device memory between passes (16 B/pixel in HBM; nothing but the 4 B/pixel preview crosses PCIe),
 * each pass adds the next n_samples samples of every pixel - the same samples the one-shot render takes, so the finished
 * frame equals rt_render_path's up to the order of the float additions - and optionally resolves a preview exposed for the
 * samples done so far (sum / done; sqrt; clamp; truncate).  Works on device groups (rt_create_multi) as rt_render_path does.
 * One progressive frame per context; rt_progressive_begin copies the scene records.  params->sample_begin/end select the
 * range the frame covers ((0,0) = [0, spp)).  rt_progressive_pass with n_samples = 0 only resolves; once the range is
 * used up further passes render nothing.  samples_done (optional) receives the samples per pixel accumulated so far;
 * rt_stats describes the pass.  Blocking. */
int32_t rt_progressive_begin(rt_ctx* ctx, const rt_scene* scene, const rt_camera* camera,
                             const rt_path_params* params, int32_t width, int32_t height);
int32_t rt_progressive_pass(rt_ctx* ctx, int32_t n_samples, const rt_image* preview, int32_t* samples_done);
int32_t rt_progressive_end(rt_ctx* ctx);

/* ---- device-buffer entry points (bench / multi-GPU composition) ----------------------- */
/* Same kernels, but outputs stay in device memory the caller owns (e.g. a torch tensor), so
 * per-GPU sample-range partial sums can be all-reduced (RCCL) before resolve.  'stream' is a
 * hipStream_t passed as void* (NULL = the context's own stream).  These calls enqueue work
 * and return; rt_sync() waits and collects ray counters/timing into rt_stats.
 * No counterpart in the reference (it has no device memory). */

/* accum: width*height float4 (sum R, sum G, sum B, number of samples added), OVERWRITTEN. */
int32_t rt_path_accumulate_device(rt_ctx* ctx, const rt_scene* scene, const rt_camera* camera,
                                  const rt_path_params* params, int32_t width, int32_t height,
                                  float* d_accum, void* stream);
/* rt_path_accumulate_device for callers that may not allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc).  Not capturable into a HIP graph: it waits on an event and records timing events on the stream.
 * No counterpart in the reference.  Same arguments, same result as
 * rt_path_accumulate_device, but everything the kernels need beyond the caller's buffers - the flattened scene, the ray
 * counters and the per-slice partial sums - lives in ONE fixed array that is part of the library's code object (a
 * static __device__ array; every GPU the library runs on holds its own copy from module load to process end), and this
 * call makes no hipMalloc / hipFree / hipHostMalloc and compiles nothing (it always runs the generic kernels, brute force).
 * Limits (rt_noalloc_limits): the flattened scene must fit scene_bytes (64 KiB: about 700 spheres/planes or 400
 * rectangles/triangles; larger -> RT_ERR_TOO_LARGE); the partial sums use at most partial_bytes (32 MiB): an image too large
 * for several slices is rendered with fewer, or written straight into d_accum - any width*height the other entry point takes.
 * RT_OPT_ACCEL must be RT_ACCEL_NONE (else RT_ERR_UNSUPPORTED).
 * All calls on one GPU - from any context, any stream - share that GPU's array, which keeps the last call's contents.  A
 * call first waits (on the host) for the previous call's kernels on that GPU, whichever context and stream issued them (one
 * guard event per GPU), so consecutive calls are safe; two host threads ENTERING the call for one GPU at the same time still
 * race: one at a time per GPU. */
int32_t rt_path_accumulate_device_noalloc(rt_ctx* ctx, const rt_scene* scene, const rt_camera* camera,
                                          const rt_path_params* params, int32_t width, int32_t height,
                                          float* d_accum, void* stream);
/* The fixed sizes above, in bytes (either pointer may be NULL). */
int32_t rt_noalloc_limits(int32_t* scene_bytes, int32_t* partial_bytes);
/* color: width*height float4 (R,G,B,1), OVERWRITTEN. */
int32_t rt_whitted_device(rt_ctx* ctx, const rt_scene* scene, const rt_camera* camera,
                          const rt_whitted_params* params, int32_t width, int32_t height,
                          float* d_color, void* stream);
/* The tail of Render: c = accum.rgb / divisor; gamma ? sqrt(c) : c; clamp to [0,1];
 * truncate(c*255) -> B,G,R,255 (03 Main.cpp:352-367; 01 Main.cpp:351-356 with gamma=0, divisor=1).
 * d_rgb (optional) receives the linear float RGB (3 floats per pixel). */
int32_t rt_resolve_device(rt_ctx* ctx, const float* d_accum, int32_t width, int32_t height,
                          float divisor, int32_t gamma, uint8_t* d_bgra, int32_t bgra_stride_bytes,
                          float* d_rgb, void* stream);
int32_t rt_sync(rt_ctx* ctx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355_H */
