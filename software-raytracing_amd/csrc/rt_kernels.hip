// rt_kernels.hip - the tracing kernels (gfx950).  One thread per pixel(-sample slice); scene
// primitives are wave-uniform and come through scalar loads; no MFMA (FP32 VALU-bound path).
#include "rt_device.h"
#include "rt_strict.h"
#include "rt_kernels.h"

namespace rt {

// =================================================================================================
// K3/K4: path tracing.  Render + TraceTay of 03 (Main.cpp:310-435) and 04 (Main.cpp:324-502).
// The recursion out = att * child (+ emitted) is a product chain, so it runs as a loop carrying
// the throughput; a lane whose path ended takes the next sample of its pixel (regeneration),
// and a wave regenerates only when enough of its lanes are idle (ballot), so lanes stay busy
// without paying the camera code on every iteration.
// =================================================================================================
// Per-lane path state of the bounce loop.
struct PathLane {
    f3 o, d, thr, acc;
    uint32_t skey;
    int depth, s_next, n_started;
    unsigned int rays;
    bool alive;
};
__device__ __forceinline__ PathLane lane_init(int s_next) {
    PathLane L;
    L.o = F3(0, 0, 0); L.d = F3(0, 0, -1); L.thr = F3(0, 0, 0); L.acc = F3(0, 0, 0);
    L.skey = 0; L.depth = 0; L.s_next = s_next; L.n_started = 0; L.rays = 0; L.alive = false;
    return L;
}

// Path regeneration: idle lanes ('want') take the next sample of their pixel once enough of the wave is idle.
__device__ __forceinline__ void regenerate(const PathArgs& A, int px, int py, uint32_t pkey, bool want, unsigned long long m_want, unsigned long long m_alive, PathLane& L) {
    if (m_want && (m_alive == 0ULL || __popcll(m_want) >= A.regen_threshold)) {
        if (want) {   // Render: one primary ray (03 Main.cpp:331-341, 04 Main.cpp:359-369)
            L.skey = rng_sample_key(pkey, (uint32_t)L.s_next);
            float bx = 0.5f, by = 0.5f;
            if (A.jitter) rng_pair(L.skey, 0U, bx, by);
            float x = fmaf((float)px + bx, A.inv_w2, -1.0f) * A.kx;
            float y = -fmaf((float)py + by, A.inv_h2, -1.0f) * A.ky;
            f3 v = F3(x, y, -1.0f);
            L.d = normalize(F3(dot(F3(A.r[0], A.r[1], A.r[2]), v), dot(F3(A.r[3], A.r[4], A.r[5]), v), dot(F3(A.r[6], A.r[7], A.r[8]), v)));
            L.o = F3(A.px, A.py, A.pz);
            L.thr = F3(1, 1, 1);
            L.depth = 0;
            L.alive = true;
            ++L.s_next; ++L.n_started;
        }
    }
}
// wave-level state of the bounce loop: which lanes are idle with samples left, which are tracing
#define RT_WAVE_STATE(L, s_end) \
    const bool want = !(L).alive && (L).s_next < (s_end); \
    const unsigned long long m_want = __builtin_amdgcn_ballot_w64(want), m_alive = __builtin_amdgcn_ballot_w64((L).alive)

// One bounce after the scene query: miss / emit / scatter (TraceTay of 03 Main.cpp:392-435, 04 Main.cpp:447-502).
// hit = SceneMem address of the winning slot (0 = miss), t = its distance.
template <int VARIANT, bool LDS>
__device__ __forceinline__ void shade(const SceneMem<LDS> M, const PathArgs& A, uint32_t mats0, uint32_t hit, float t, PathLane& L) {
    if (hit == 0U) {
        if (VARIANT == 3) {   // sky: Mix(white, (0.5,0.7,1), 0.5*(d.y+1)) (03 Main.cpp:430-432)
            float h = 0.5f * (L.d.y + 1.0f);
            L.acc = L.acc + L.thr * F3(fmaf(-0.5f, h, 1.0f), fmaf(-0.3f, h, 1.0f), 1.0f);
        }
        L.alive = false;
    } else {
        // hit record (03 Sphere.hpp:57-73; 04 Scene/Sphere.hpp:58-74, Plane.hpp:55-66, Rectangle.hpp:79-92), one
        // formula for every primitive type (slot layout, rt_device.h): n = (p - c)*k + m, turned against the ray
        float4 sa = M.ld4(hit + SLOT_A), sb = M.ld4(hit + SLOT_B), sc = M.ld4(hit + SLOT_C);
        L.o = madd(L.d, t, L.o);                       // the hit point is the next origin; nothing below needs the old one
        f3 pc = L.o - F3(sa);
        // spheres: Normalize(p - c) (k = +-1; also right when the hit was reported at tMax, off the surface); others: k = 0
        float k = sa.w * rsq_fast(fmaf(pc.z, pc.z, fmaf(pc.y, pc.y, fmaf(pc.x, pc.x, 1e-30f))));
        f3 n = madd(pc, k, F3(sb));
        float facing = dot(L.d, madd(pc, k, F3(sc)));
        bool back = facing > 0.0f;
        // n = back ? -n : n through the sign bit of (0 - facing): set exactly when facing > 0
        uint32_t flip = __float_as_uint(0.0f - facing) & 0x80000000U;
        n = F3(__uint_as_float(__float_as_uint(n.x) ^ flip), __uint_as_float(__float_as_uint(n.y) ^ flip), __uint_as_float(__float_as_uint(n.z) ^ flip));
        const uint32_t moff = mats0 + __float_as_uint(sb.w);
        float4 m0 = M.ld4(moff), m2 = M.ld4(moff + 32U);
        int mtype = __float_as_int(m0.w);
        if (mtype == MAT_DIFFUSE || mtype == MAT_METAL || mtype == MAT_DIELECTRIC) {
            // one draw per bounce serves whichever material was hit: (fi, theta) halves for the hemisphere/cone
            // samplers, the top 24 bits for the dielectric's reflect-or-refract choice
            const uint32_t r = rng_bits(L.skey, 4U * ((uint32_t)L.depth + 1U));
            const float dn = dot(L.d, n);
            const f3 refl = madd(n, -2.0f * dn, L.d);          // math::Reflect (Math.hpp:291-294)
            f3 nd;
            if (mtype == MAT_DIELECTRIC) {                   // 03 Materials.hpp:233-265, 04 Refractive.hpp:80-104
                float eta = back ? m0.x : m2.z;              // m0.x = 1/refractionIndex (host-precomputed)
                float q = 1.0f - fmaxf(-dn, 0.0f), q2 = q * q;
                float refl_p = fmaf(1.0f - m2.w, q2 * q2 * q, m2.w);
                float u = (float)(r >> 8) * (1.0f / 16777216.0f);
                // math::Refract, reflect-on-TIR form (Math.hpp:306-312)
                float k = fmaf(-eta * eta, fmaf(-dn, dn, 1.0f), 1.0f);
                float s = fmaf(eta, dn, sqrt_fast(fmaxf(k, 0.0f)));
                f3 refr = madd(n, -s, L.d * eta);
                nd = sel((refl_p < u) & !(k < 0.0f), refr, refl);
            } else {
                f3 axis = sel(mtype == MAT_METAL, refl, n);
                nd = axis;
                if (m2.y >= 0.0f) {
                    // RndHemisphereVec (03 Utils.h:64-88) / RndHemisphereVec3 (04 Utils.h:132-157)
                    float u_fi = (float)(r & 0xFFFFU) * (1.0f / 65536.0f), u_th = (float)(r >> 16) * (1.0f / 65536.0f);
                    // b = Normalize(Cross(dir, dir + 0.01)) = Normalize(Cross(dir, (1,1,1))): the 0.01 and dir x dir cancel;
                    // c = Cross(dir, b) is already unit (dir, b orthonormal), so the reference's second Normalize is a no-op
                    f3 b = F3(axis.y - axis.z, axis.z - axis.x, axis.x - axis.y);
                    float bl2 = dot(b, b);
                    bool degenerate = !(bl2 > 0.0f);
                    b = b * (degenerate ? 0.0f : rsq_fast(bl2));      // Normalize(0) = 0: the frame collapses for dir parallel to (1,1,1)
                    f3 c = cross(axis, b);
                    // fi = U*360/57.2958; cos from the hardware (input in revolutions), sin = +-sqrt(1 - cos^2) so that
                    // cos^2 + sin^2 = 1 to the last bit: v_sin/v_cos are each good to ~1e-5 only, and the sphere test
                    // amplifies an error in |L.d| by (distance/radius)^2
                    const float fi_rev = (float)((360.0 / 57.2958f) / 6.283185307179586);   // fi/(2*pi)
                    float x_rev = u_fi * fi_rev;
                    float cf = __builtin_amdgcn_cosf(x_rev);
                    float sf = sqrt_fast(fmaf(-cf, cf, 1.0f));
                    sf = __uint_as_float(__float_as_uint(sf) | (__float_as_uint(0.5f - x_rev) & 0x80000000U));   // negative beyond half a turn
                    float ct, st;
                    if (VARIANT == 3) {   // theta = U(-1,1)*thetaMax/57.2958
                        float th_rev = (2.0f * u_th - 1.0f) * m2.y * 0.15915494309189535f;
                        ct = __builtin_amdgcn_cosf(th_rev);
                        st = sqrt_fast(fmaf(-ct, ct, 1.0f));
                        st = __uint_as_float(__float_as_uint(st) | (__float_as_uint(th_rev) & 0x80000000U));      // |theta| <= 90 deg: sin has theta's sign
                    } else {              // theta = acos(U(cos(thetaMax/57.2958), 1))
                        ct = fmaf(1.0f - m2.y, u_th, m2.y);
                        st = sqrt_fast(fmaf(-ct, ct, 1.0f));   // ct <= 1 exactly (u_th < 1), so the fused 1 - ct*ct is never negative
                    }
                    nd = madd(madd(c, sf, b * cf), st, axis * ct);
                    // Ray's constructor normalises (Common/Ray.hpp:33-34); nd is unit by construction unless the frame
                    // collapsed, where it is axis*cos(theta)
                    if (degenerate) nd = normalize(nd);
                }
            }
            L.d = nd;   // unit: n is unit (slot layout), Reflect/Refract keep the length, the sampled direction is unit by construction
            if (mtype != MAT_DIELECTRIC) {
                f3 att = F3(m0);
                if (VARIANT == 4) att = att * fmaxf(dot(L.d, n), 0.0f);      // 04 Diffuse.hpp:77-78, Metal.hpp:83-84
                L.thr = L.thr * att;
            }
            ++L.depth;
        } else {
            if (VARIANT == 4 && mtype == MAT_LIGHT && !back) L.acc = L.acc + L.thr * F3(m0);   // Light.hpp:47-54, Main.cpp:480-483
            L.alive = false;
        }
        if (L.depth > A.max_depth) L.alive = false;   // 03 Main.cpp:383-386, 04 Main.cpp:438-441
    }
}

// Small scenes (blob <= 48 KB, LDS = true): every block stages the whole blob in LDS once.  LDS = false: the blob is
// read from global memory with scalar loads (debug flag 2; A/B reference for the other two kernels).
template <int VARIANT, bool LDS>
__global__ __launch_bounds__(256) void k_path(DevScene SD, PathArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    extern __shared__ float4 smem[];
    const char* base = reinterpret_cast<const char*>(SD.blob);
    if (LDS) {   // primitive operands then arrive in VGPRs (ds_read), not SGPRs
        for (int i = threadIdx.x; i < SD.total_f4; i += 256) smem[i] = SD.blob[i];
        __syncthreads();
        base = reinterpret_cast<const char*>(smem);
    }
    const int4* __restrict__ runs = reinterpret_cast<const int4*>(SD.blob + SD.off_sruns);   // always global: scalar loads
    const SceneMem<LDS> M{base};
    const uint32_t slots0 = M.at((uint32_t)SD.off_slots * 16U), mats0 = M.at(0U);
    int px, py;
    bool inside = pixel_of_thread(A.width, A.height, px, py);
    const int slice = blockIdx.y;
    int s_first = A.sample_begin + slice * A.slice_len;
    const int s_end = min(s_first + A.slice_len, A.sample_end);
    if (!inside) s_first = s_end;
    const uint32_t pixel = (uint32_t)(py * A.width + px);
    const uint32_t pkey = rng_pixel_key(A.seed, pixel);
    PathLane L = lane_init(s_first);

    for (;;) {
        RT_WAVE_STATE(L, s_end);
        if ((m_want | m_alive) == 0ULL) break;
        regenerate(A, px, py, pkey, want, m_want, m_alive, L);
        if (L.alive) {
            // TraceTay: one scene query (03 Main.cpp:392, 04 Main.cpp:447)
            ++L.rays;
            float t;
            uint32_t hit = closest_hit_slots<LDS>(M, runs, SD.n_sruns, slots0, L.o, L.d, A.t_min, A.t_min_pos, A.t_max, t);
            shade<VARIANT, LDS>(M, A, mats0, hit, t, L);
        }
    }
    if (inside) {
        // debug flag 4: report the lane's scene-query count in w instead of its sample count (parity diagnostics)
        out[(size_t)slice * A.width * A.height + pixel] = make_float4(L.acc.x, L.acc.y, L.acc.z, (A.debug_flags & 4) ? (float)L.rays : (float)L.n_started);
    }
    wave_add_u64(counters + 2 * ((blockIdx.x + blockIdx.y) % RT_COUNTER_BANKS), L.rays);
}

// Large scenes (blob > 48 KB): the slot section is walked in pieces of <= RT_TILE_BYTES that the block's four waves
// stage through LDS together (one global read per block instead of one per wave, and VGPR operands for the tests), so
// the bounce loop runs in lockstep per BLOCK: regenerate per wave, then every chunk between two barriers, then shade.
// The winner is remembered as a blob offset (its chunk is gone from LDS by shading time) and shaded from global memory.
template <int VARIANT>
__global__ __launch_bounds__(256) void k_path_tiled(DevScene SD, PathArgs A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    extern __shared__ float4 smem[];
    const int4* __restrict__ runs = reinterpret_cast<const int4*>(SD.blob + SD.off_sruns);
    const int4* __restrict__ chunks = reinterpret_cast<const int4*>(SD.blob + SD.off_chunks);
    const SceneMem<false> G{reinterpret_cast<const char*>(SD.blob)};
    const SceneMem<true> T{reinterpret_cast<const char*>(smem)};
    const uint32_t tile0 = T.at(0U);
    int px, py;
    bool inside = pixel_of_thread(A.width, A.height, px, py);
    const int slice = blockIdx.y;
    int s_first = A.sample_begin + slice * A.slice_len;
    const int s_end = min(s_first + A.slice_len, A.sample_end);
    if (!inside) s_first = s_end;
    const uint32_t pixel = (uint32_t)(py * A.width + px);
    const uint32_t pkey = rng_pixel_key(A.seed, pixel);
    PathLane L = lane_init(s_first);

    for (;;) {
        {
            RT_WAVE_STATE(L, s_end);
            regenerate(A, px, py, pkey, want, m_want, m_alive, L);
        }
        if (!__syncthreads_or(L.alive)) break;      // block-uniform: a finished wave keeps staging chunks for the others
        float closest = A.t_max;
        uint32_t best = 0U;
        const f3 inv_d = F3(rcp_fast(L.d.x), rcp_fast(L.d.y), rcp_fast(L.d.z));
        for (int c = 0; c < SD.n_chunks; ++c) {
            const int4 ch = chunks[c];
            const float4* __restrict__ src = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(SD.blob) + ch.x);
            if (c) __syncthreads();                 // everyone is done with the previous chunk
            for (int i = threadIdx.x; i < (ch.y >> 4); i += 256) smem[i] = src[i];
            __syncthreads();
            if (L.alive) {
                uint32_t b = 0U;
                closest_hit_runs<true>(T, runs + ch.z, ch.w, tile0, L.o, L.d, inv_d, A.t_min, A.t_min_pos, closest, b);
                if (b) best = b - tile0 + (uint32_t)ch.x;
            }
        }
        if (L.alive) {
            ++L.rays;
            shade<VARIANT, false>(G, A, 0U, best, closest, L);
        }
    }
    if (inside) out[(size_t)slice * A.width * A.height + pixel] = make_float4(L.acc.x, L.acc.y, L.acc.z, (A.debug_flags & 4) ? (float)L.rays : (float)L.n_started);
    wave_add_u64(counters + 2 * ((blockIdx.x + blockIdx.y) % RT_COUNTER_BANKS), L.rays);
}

template <int VARIANT>
static void launch_path_v(const DevScene& S, const PathArgs& A, float4* out, int n_slices, unsigned long long* counters, hipStream_t st) {
    dim3 grid(((A.width + 15) / 16) * ((A.height + 15) / 16), n_slices);
    size_t bytes = (size_t)S.total_f4 * sizeof(float4);
    if (A.debug_flags & 2)                   hipLaunchKernelGGL((k_path<VARIANT, false>), grid, dim3(256), 0, st, S, A, out, counters);   // A/B reference: scalar loads from the global blob
    else if (bytes <= RT_LDS_SCENE_MAX_BYTES) hipLaunchKernelGGL((k_path<VARIANT, true>), grid, dim3(256), bytes, st, S, A, out, counters);  // small scene: LDS-resident
    else                                     hipLaunchKernelGGL((k_path_tiled<VARIANT>), grid, dim3(256), RT_TILE_BYTES, st, S, A, out, counters);   // large scene: LDS-tiled
}
void launch_path(int variant, const DevScene& S, const PathArgs& A, float4* out, int n_slices, unsigned long long* counters, hipStream_t st) {
    if (variant == 3) launch_path_v<3>(S, A, out, n_slices, counters, st);
    else              launch_path_v<4>(S, A, out, n_slices, counters, st);
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
