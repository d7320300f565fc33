// rt_jit.h - scene-specialised path kernels (hiprtc).  See rt_jit.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>

namespace rt {
constexpr int RT_JIT_WHITTED_SOFT = 2;   // 'variant' of the 02 kernel (3 and 4 are the path variants)
constexpr int RT_JIT_WHITTED_HARD = 1;   // 'variant' of the 01 kernel (one shadow ray per shaded hit and light, strict arithmetic throughout)
// Returns the kernel for this (variant, slot section) or nullptr when specialisation is unavailable (hiprtc missing, compile
// error, scene too large); 'note' then says why.  Compiled kernels are cached per process by the hash of their source.
// Policy: whether this render should use (and, if need be, compile) the specialised kernel; counts sightings of the scene.
bool jit_wanted(int variant, const std::vector<float4>& slots, const std::vector<int4>& sruns, double job_samples, bool force, std::string& note);
// tmin / tmax: the range of the path query (literals of the generated code; ignored by the 02 kernel, whose shadow rays carry their own)
// wruns / wgeom (Whitted variants): the IEEE-exact layout of rt_strict.h (runs {type, first float4 of geom, count, first primitive}; geom): the closest-hit and
// any-hit queries of the tree walk are generated from it too, as calls of the strict tests with literal operands (same operations, bit for bit)
hipFunction_t jit_path_kernel(int variant, const std::vector<float4>& slots, const std::vector<int4>& sruns, float tmin, float tmax, int form, std::string& note, hipFunction_t* second = nullptr,
                              const std::vector<int4>* wruns = nullptr, const std::vector<float4>* wgeom = nullptr);
// second (02 only): the module's other kernel, stage 2 of the wavefront form (k_whitted_shadow_jit)
// form: which body the kernel wraps - RT_FORM_REGEN path_body (per-lane regeneration), RT_FORM_QUEUE path_queue_body (workgroup-level
// ray-queue compaction), RT_FORM_ITEMS path_items_body (persistent waves, work counter, lanes decoupled from pixels)
// RT_FORM_SORTED path_sorted_body (the workgroup sorts its paths by the outcome of the scene query)
enum { RT_FORM_REGEN = 0, RT_FORM_QUEUE = 1, RT_FORM_ITEMS = 2, RT_FORM_SORTED = 3,
       RT_FORM_REF = 8, RT_FORM_NORM = 16, RT_FORM_POINT = 32 };      // flag, with RT_FORM_REGEN only: the reference's arithmetic for rays and spheres (RT_ARITH_REFERENCE; the source defines RT_REF_RAYS true); RT_FORM_NORM: flag, any form: PathArgs::norm_dirs as a literal (RT_JIT_NORM_DIRS); RT_FORM_POINT: likewise PathArgs::strict_point (RT_JIT_STRICT_POINT)
}
