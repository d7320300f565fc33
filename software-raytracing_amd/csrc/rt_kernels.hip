// rt_kernels.hip - the tracing kernels (gfx950).  One thread per pixel(-sample slice); scene
// primitives are wave-uniform and come through scalar loads; no MFMA (FP32 VALU-bound path).
#include "rt_device.h"
#include "rt_strict.h"
#include "rt_kernels.h"
#include "rt_path_device.h"
#include "rt_bvh_device.h"

namespace rt {

// =================================================================================================
// K3/K4: path tracing (device code in rt_path_device.h).
// =================================================================================================
template <int VARIANT, bool LDS>
__global__ __launch_bounds__(256) void k_path(DevScene SD, PathArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    path_body<VARIANT, LDS>(SD, A, out, counters);
}
template <int VARIANT>
__global__ __launch_bounds__(256) void k_path_tiled(DevScene SD, PathArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    path_tiled_body<VARIANT>(SD, A, out, counters);
}

template <int VARIANT>
static void launch_path_v(const DevScene& S, const PathArgs& A, float4* out, int n_slices, unsigned long long* counters, hipStream_t st, hipFunction_t jit) {
    dim3 grid(((A.width + 15) / 16) * ((A.height + 15) / 16), n_slices);
    size_t bytes = (size_t)S.total_f4 * sizeof(float4);
    if (jit && bytes <= RT_LDS_SCENE_MAX_BYTES) {   // scene-specialised k_path<VARIANT, true> (rt_jit.hip)
        DevScene s = S; PathArgs a = A;
        void* params[] = {&s, &a, &out, &counters};
        (void)hipModuleLaunchKernel(jit, grid.x, grid.y, 1, 256, 1, 1, (unsigned)bytes, st, params, nullptr);
        return;
    }
    if (A.debug_flags & 2)                   hipLaunchKernelGGL((k_path<VARIANT, false>), grid, dim3(256), 0, st, S, A, out, counters);   // A/B reference: scalar loads from the global blob
    else if (bytes <= RT_LDS_SCENE_MAX_BYTES) hipLaunchKernelGGL((k_path<VARIANT, true>), grid, dim3(256), bytes, st, S, A, out, counters);  // small scene: LDS-resident
    else                                     hipLaunchKernelGGL((k_path_tiled<VARIANT>), grid, dim3(256), RT_TILE_BYTES, st, S, A, out, counters);   // large scene: LDS-tiled
}
void launch_path(int variant, const DevScene& S, const PathArgs& A, float4* out, int n_slices, unsigned long long* counters, hipStream_t st, hipFunction_t jit) {
    if (variant == 3) launch_path_v<3>(S, A, out, n_slices, counters, st, jit);
    else              launch_path_v<4>(S, A, out, n_slices, counters, st, jit);
}

// Optional BVH walk (rt_set_option(RT_OPT_ACCEL, RT_ACCEL_BVH)): same bounce loop, the scene query walks a tree.
template <int VARIANT>
__global__ __launch_bounds__(256) void k_path_bvh(DevScene SD, PathArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    path_bvh_body<VARIANT>(SD, A, out, counters);
}
void launch_path_bvh(int variant, const DevScene& S, const PathArgs& A, float4* out, int n_slices, unsigned long long* counters, hipStream_t st) {
    dim3 grid(((A.width + 15) / 16) * ((A.height + 15) / 16), n_slices);
    // the walk is bound by the latency of dependent fetches, so resident waves matter: the per-lane stack takes only what this tree needs
    const size_t lds = (size_t)(S.bvh_depth > 0 ? S.bvh_depth : 1) * 256 * sizeof(uint32_t);
    if (variant == 3) hipLaunchKernelGGL((k_path_bvh<3>), grid, dim3(256), lds, st, S, A, out, counters);
    else              hipLaunchKernelGGL((k_path_bvh<4>), grid, dim3(256), lds, st, S, A, out, counters);
}

// Sum the per-slice partial images in a fixed order (deterministic), out = float4(sum rgb, samples).
__global__ __launch_bounds__(256) void k_sum_slices(const float4* __restrict__ part, int n_pixels, int n_slices, float4* __restrict__ out) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pixels) return;
    float4 a = part[i];
    for (int s = 1; s < n_slices; ++s) { float4 b = part[(size_t)s * n_pixels + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    out[i] = a;
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
#define WBLOCK 128
template <bool SOFT>
// 8 waves/SIMD (64 VGPRs, a few cold values spilled): this kernel is bound by dependent-issue latency, not registers:
// measured 02 4K/256 samples 10.5 ms at 4 waves, 9.3 at 5, 8.7 at 6, 8.3 at 8
__global__ __launch_bounds__(WBLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_whitted(DevScene SD, WhittedArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    const Scn S = scn_over(SD, SD.blob);
    // slot k (k = depth of the parked child, 1..max_depth+1): o(3) d(3) weight node, one column per thread; sized by the
    // launch to max_depth + 2 slots (4 KB each), so the default depth 4 leaves room for 6 blocks per CU
    extern __shared__ float stk_raw[];
    float (*stk)[8][WBLOCK] = reinterpret_cast<float (*)[8][WBLOCK]>(stk_raw);
    // 02: the shadow samples (almost all of its rays) run on the slot layout staged in LDS behind the stack, with the
    // fast tests of the path kernels (see fastshadow in rt_strict.h for why that is safe); A.shadow_lds = 0 when the
    // scene is too large for that
    const SceneMem<true> T{reinterpret_cast<const char*>(stk_raw)};
    const uint32_t sh_slots = T.at((uint32_t)A.stack_bytes);
    const int4* __restrict__ sruns = reinterpret_cast<const int4*>(SD.blob + SD.off_sruns);
    if (SOFT && A.shadow_lds) {
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<char*>(stk_raw) + A.stack_bytes);
        const float4* src = SD.blob + SD.off_slots;
        for (int i = threadIdx.x; i < (SD.slot_bytes >> 4); i += WBLOCK) dst[i] = src[i];
        __syncthreads();
    }
    const int tid = threadIdx.x;
    // 16x8 pixels per 128-thread block, 8x8 per wave
    int tiles_x = (A.width + 15) >> 4;
    int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    int px = tx * 16 + (tid >> 6) * 8 + (tid & 7), py = ty * 8 + ((tid & 63) >> 3);
    bool inside = px < A.width && py < A.height;
    const uint32_t pixel = (uint32_t)(py * A.width + px);
    const uint32_t skey = rng_sample_key(rng_pixel_key(A.seed, pixel), 0U);

    f3 o = F3(0, 0, 0), d = strict::primary_dir(px, py, A.w, A.h, A.tan_half);     // 01 Main.cpp:340-344
    float weight = 1.0f, tmin = 0.0f;
    int depth = 0;
    uint32_t node = 1U, pending = 0U;
    f3 acc = F3(0, 0, 0);
    unsigned int rays = 0, srays = 0;
    bool active = inside;
    const f3 sky = F3(0.2f, 0.7f, 0.8f);

    while (__any(active)) {
        if (active) {
            bool done_node = true;   // pop after this node unless a child continues in place
            float t; int idx = -1;
            if (depth <= A.max_depth) { ++rays; idx = strict::closest_hit(S, o, d, tmin, 1000.0f, t); }          // :386-403
            if (idx < 0) {
                acc = madd(sky, weight, acc);                                                            // :489-493
            } else {
                int4 inf = S.info[idx];
                f3 p = strict::add(o, strict::muls(d, t));                                               // :414
                f3 n = (inf.x == PRIM_SPHERE) ? strict::normalize(strict::sub(p, F3(S.geom[inf.y]))) : F3(S.geom[inf.y]);   // Sphere.hpp:48, Plane.hpp:48
                float4 m0 = S.mats[inf.z * 3], m1 = S.mats[inf.z * 3 + 1], m2 = S.mats[inf.z * 3 + 2];
                float p2s = m1.z, r2r = m1.w;
                float wp = fmaxf(1.0f - p2s, 0.0f), wr = fmaxf(1.0f - r2r, 0.0f);
                if (p2s > 0.0f) {                                                                        // :421-447
                    f3 diffuse = F3(0, 0, 0), specular = F3(0, 0, 0);
                    for (int li = 0; li < S.n_lights; ++li) {
                        float4 l0 = S.lights[2 * li], l1 = S.lights[2 * li + 1];
                        f3 lp = F3(l0);
                        f3 to = strict::sub(lp, p);
                        f3 tod = strict::normalize(to);
                        float len = strict::length(to);
                        float intensity = 1.0f;
                        bool skip = false;
                        if (!SOFT) {                                                                     // 01 :435-436
                            if (depth + 1 <= A.max_depth) { ++srays; skip = strict::any_hit(S, p, tod, 0.01f, len); }
                        } else if (depth + 1 <= A.max_depth) {                                           // 02 :452-467, :539-566
                            float per = 1.0f / (float)A.shadow_samples;
                            float tmax_s = len - l0.w;
                            for (int i = 0; i < A.shadow_samples; ++i) {
                                uint32_t c = ((node * (uint32_t)S.n_lights + (uint32_t)li) * (uint32_t)A.shadow_samples + (uint32_t)i) * 4U;
                                float u0, u1, u2, u3;
                                rng_pair(skey, c, u0, u1); rng_pair(skey, c + 2U, u2, u3);
                                f3 sd = fastshadow::to_light_disk(p, lp, tod, l0.w, u0, u1, u2, u3);
                                ++srays;
                                bool occluded;
                                if (A.shadow_lds) {
                                    float closest = tmax_s; uint32_t best = 0U;
                                    closest_hit_runs<true, true>(T, sruns, SD.n_sruns, sh_slots, p, sd, F3(rcp_fast(sd.x), rcp_fast(sd.y), rcp_fast(sd.z)), 0.01f, 0.01f, closest, best, SD.srun_pack[0], SD.srun_pack[1]);
                                    occluded = best != 0U;
                                } else occluded = fastshadow::any_hit(S, p, sd, 0.01f, tmax_s);
                                if (occluded) intensity -= per;
                            }
                        }
                        if (skip) continue;   // 01: an occluded light is skipped entirely
                        float dd = fmaxf(0.0f, strict::dot(tod, n));
                        float sdot = fmaxf(0.0f, strict::dot(strict::reflect(-tod, n), -d));
                        float sp = powf(sdot, m1.y);
                        diffuse = madd(F3(l1), dd * intensity, diffuse);
                        specular = madd(F3(l1), sp * intensity, specular);
                    }
                    f3 primary = diffuse * F3(m0) + specular * m1.x;                                     // :446
                    acc = madd(primary, weight * (1.0f - wp), acc);
                }
                if (p2s < 1.0f) {                                                                        // :450-481
                    bool do_refl = r2r > 0.0f, do_refr = r2r < 1.0f;
                    float w_refl = weight * wp * (1.0f - wr), w_refr = weight * wp * wr;
                    f3 d_refl = strict::normalize(strict::reflect(d, n)), d_refr = d;
                    if (do_refr) {                                                                       // :467-475
                        float eta = m2.x;
                        f3 nn = n;
                        if (strict::dot(n, -d) < 0.0f) { nn = -n; eta = 1.0f / eta; }
                        d_refr = strict::normalize(strict::refract(d, nn, eta));
                    }
                    int cd = depth + 1;
                    if (do_refl && do_refr) {   // park the refraction child in the slot of its depth
                        stk[cd][0][tid] = p.x; stk[cd][1][tid] = p.y; stk[cd][2][tid] = p.z;
                        stk[cd][3][tid] = d_refr.x; stk[cd][4][tid] = d_refr.y; stk[cd][5][tid] = d_refr.z;
                        stk[cd][6][tid] = w_refr; stk[cd][7][tid] = __uint_as_float(node * 2U + 1U);
                        pending |= 1U << cd;
                    }
                    if (do_refl || do_refr) {
                        o = p; tmin = 0.001f; depth = cd;
                        if (do_refl) { d = d_refl; weight = w_refl; node = node * 2U; }
                        else         { d = d_refr; weight = w_refr; node = node * 2U + 1U; }
                        done_node = false;
                    }
                }
            }
            if (done_node) {
                if (pending == 0U) active = false;
                else {
                    int k = 31 - __clz((int)pending);
                    pending &= ~(1U << k);
                    o = F3(stk[k][0][tid], stk[k][1][tid], stk[k][2][tid]);
                    d = F3(stk[k][3][tid], stk[k][4][tid], stk[k][5][tid]);
                    weight = stk[k][6][tid]; node = __float_as_uint(stk[k][7][tid]);
                    depth = k; tmin = 0.001f;
                }
            }
        }
    }
    if (inside) out[pixel] = make_float4(acc.x, acc.y, acc.z, 1.0f);
    wave_add_u64(counters + 2 * (blockIdx.x % RT_COUNTER_BANKS), rays);
    wave_add_u64(counters + 2 * (blockIdx.x % RT_COUNTER_BANKS) + 1, srays);
}

void launch_whitted(bool soft, const DevScene& S, const WhittedArgs& A, float4* out, unsigned long long* counters, hipStream_t st) {
    dim3 grid(((A.width + 15) / 16) * ((A.height + 7) / 8));
    WhittedArgs B = A;
    // the stack only ever holds the refraction child of a hit that ALSO reflects (0 < reflectToRefract < 1 on a material with
    // primaryToSecondary < 1): none of the reference's scenes has one, and then no LDS is spent on it (more blocks per CU)
    B.stack_bytes = S.whitted_branches ? (A.max_depth + 2) * 8 * WBLOCK * (int)sizeof(float) : 0;
    B.shadow_lds = soft && S.slot_bytes > 0 && S.slot_bytes <= RT_TILE_BYTES && S.srun_packed;
    size_t lds = (size_t)B.stack_bytes + (B.shadow_lds ? (size_t)S.slot_bytes + 256 : 0);   // + one slot of slack for the read-ahead of wide slots
    if (soft) hipLaunchKernelGGL(k_whitted<true>, grid, dim3(WBLOCK), lds, st, S, B, out, counters);
    else      hipLaunchKernelGGL(k_whitted<false>, grid, dim3(WBLOCK), lds, st, S, B, out, counters);
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
