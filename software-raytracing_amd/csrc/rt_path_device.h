// rt_path_device.h - device code of the path-tracing kernels (03/04): per-lane state, regeneration, shading and the
// two kernel bodies.  Included by rt_kernels.hip and, as text, by the scene-specialised hiprtc builds (rt_jit.hip),
// so it must compile without any system include.
#pragma once
#include "rt_args.h"
#include "rt_device.h"

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
    uint32_t skey, rkey;      // rkey = skey + (4*(depth+1) + 1)*0x9E3779B9: the argument of the bounce's draw, kept by adding instead of multiplying
    int depth, s_next, n_started;
    unsigned int rays;
    bool alive;
    uint32_t pl, acc0;        // item form only: the path's pixel within its unit, and the LDS address of the unit's fixed-point sums (wave-uniform)
};
__device__ __forceinline__ PathLane lane_init(int s_next) {
    PathLane L;
    L.o = F3(0, 0, 0); L.d = F3(0, 0, -1); L.thr = F3(0, 0, 0); L.acc = F3(0, 0, 0);
    L.skey = 0; L.rkey = 0; L.depth = 0; L.s_next = s_next; L.n_started = 0; L.rays = 0; L.alive = false; L.pl = 0; L.acc0 = 0;
    return L;
}

// Path regeneration: idle lanes ('want') take the next sample of their pixel once enough of the wave is idle.
// REF: the reference's arithmetic for rays (RT_ARITH_REFERENCE; rt_device.h strict_sphere_roots) - directions normalised as Ray's constructor normalises them.
template <bool REF = RT_REF_RAYS>
__device__ __forceinline__ void regenerate(const PathArgs& A, int px, int py, uint32_t pkey, bool want, unsigned long long m_want, unsigned long long m_alive, PathLane& L) {
    if (m_want && (m_alive == 0ULL || __popcll(m_want) >= A.regen_threshold)) {
        if (want) {   // Render: one primary ray (03 Main.cpp:331-341, 04 Main.cpp:359-369)
            L.skey = rng_sample_key(pkey, (uint32_t)L.s_next);
            L.rkey = L.skey + 5U * 0x9E3779B9U;
            float bx = 0.5f, by = 0.5f;
            if (A.jitter) rng_pair(L.skey, 0U, bx, by);
            float x = fmaf((float)px + bx, A.inv_w2, -1.0f) * A.kx;
            float y = -fmaf((float)py + by, A.inv_h2, -1.0f) * A.ky;
            f3 v = F3(x, y, -1.0f);
            const f3 rv = F3(dot(F3(A.r[0], A.r[1], A.r[2]), v), dot(F3(A.r[3], A.r[4], A.r[5]), v), dot(F3(A.r[6], A.r[7], A.r[8]), v));
            L.d = REF ? strict_dir(rv) : rv * rsq_fast(dot(rv, rv));      // Normalize; |rv| >= 1 (v.z = -1, r a rotation), so its zero-vector guard cannot fire
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
    const unsigned long long m_alive = __builtin_amdgcn_ballot_w64((L).alive), \
                             m_want = ~m_alive & __builtin_amdgcn_ballot_w64((L).s_next < (s_end))   /* the compare's mask is fresh: no round trip through a VGPR */

// Radiance of a path that ends (the sky of 03, a light's front face in 04).  DEPOSIT = false: into the lane's own running sums
// (the lane owns a pixel).  DEPOSIT = true (item form): into the pixel's 32.32 fixed-point sums in LDS - integer addition is
// order-independent, so the frame does not depend on which lane traced which path.  Contributions are taken as >= 0 and < 2^32.
__device__ __forceinline__ unsigned long long to_fixed_32_32(float x) {
    const uint32_t hi = (uint32_t)x;                                           // saturating, NaN -> 0
    return ((unsigned long long)hi << 32) | (uint32_t)(__builtin_amdgcn_fractf(x) * 4294967296.0f);
}
template <bool DEPOSIT>
__device__ __forceinline__ void add_radiance(PathLane& L, f3 v) {
    if (DEPOSIT) {
        RT_LDS_AS unsigned long long* a = (RT_LDS_AS unsigned long long*)(uintptr_t)(L.acc0 + L.pl * 24U);
        __hip_atomic_fetch_add(a, to_fixed_32_32(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(a + 1, to_fixed_32_32(v.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(a + 2, to_fixed_32_32(v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else L.acc = L.acc + v;
}

// One bounce after the scene query: miss / emit / scatter (TraceTay of 03 Main.cpp:392-435, 04 Main.cpp:447-502).
// hit = SceneMem address of the winning slot (unused when miss), t = its distance.
template <int VARIANT, bool LDS, bool DEPOSIT = false, bool REF = RT_REF_RAYS>
__device__ __forceinline__ void shade(const SceneMem<LDS> M, const PathArgs& A, uint32_t mats0, bool miss, uint32_t hit, float t, PathLane& L) {
    if (miss) {
        if (VARIANT == 3) {   // sky: Mix(white, (0.5,0.7,1), 0.5*(d.y+1)) (03 Main.cpp:430-432)
            float h = 0.5f * (L.d.y + 1.0f);
            add_radiance<DEPOSIT>(L, L.thr * F3(fmaf(-0.5f, h, 1.0f), fmaf(-0.3f, h, 1.0f), 1.0f));
        }
        L.alive = false;
    } else {
        // hit record (03 Sphere.hpp:57-73; 04 Scene/Sphere.hpp:58-74, Plane.hpp:55-66, Rectangle.hpp:79-92), one
        // formula for every primitive type (slot layout, rt_device.h): n = (p - c)*k + m, turned against the ray
        float4 sa = M.ld4(hit + SLOT_A), sb = M.ld4(hit + SLOT_B), sc = M.ld4(hit + SLOT_C);
        L.o = (REF || RT_STRICT_POINT(A) || RT_NORM_DIRS(A)) ? strict_point(L.o, L.d, t) : madd(L.d, t, L.o);      // the hit point is the next origin; nothing below needs the old one
        f3 pc = L.o - F3(sa);
        // spheres: Normalize(p - c) (k = +-1; also right when the hit was reported at tMax, off the surface); others: k = 0
        float k = (REF || RT_NORM_DIRS(A)) ? (sa.w != 0.0f ? sa.w * strict_inv_length(pc) : 0.0f) : sa.w * rsq_fast(fmaf(pc.z, pc.z, fmaf(pc.y, pc.y, fmaf(pc.x, pc.x, 1e-30f))));
        f3 n = madd(pc, k, F3(sb));
        float facing = dot(L.d, madd(pc, k, F3(sc)));
        bool back = facing > 0.0f;
        n = sel(back, -n, n);      // face-forward (03 Sphere.hpp:69-73, 04 Scene/Sphere.hpp:70-74)
        const uint32_t moff = mats0 + __float_as_uint(sb.w);
        float4 m0 = M.ld4(moff), m2 = M.ld4(moff + 32U);
        int mtype = __float_as_int(m0.w);
        const bool scatters = mtype == MAT_DIFFUSE || mtype == MAT_METAL || mtype == MAT_DIELECTRIC;
        if (scatters && L.depth >= A.max_depth) {
            // the scattered ray would be traced beyond the depth limit, i.e. return black (03 Main.cpp:383-386, 04 Main.cpp:438-441), and a
            // scattering material adds nothing of its own: the path ends here, without the sampling.  In a closed scene nearly every
            // path runs to the limit and a wave's lanes reach it together (measured on 04: 63.5 of 64 lanes regenerate in the same
            // iteration), so every (max_depth + 1)-th iteration of a wave skips the whole scatter code.
            L.alive = false;
        } else if (scatters) {
            // one draw per bounce serves whichever material was hit: (fi, theta) halves for the hemisphere/cone
            // samplers, the top 24 bits for the dielectric's reflect-or-refract choice
            const uint32_t r = lowbias32(L.rkey);      // = rng_bits(L.skey, 4*(depth+1))
            L.rkey += 4U * 0x9E3779B9U;
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
                    // the uniforms are k/65536; the power-of-two scale is folded into the constants they multiply (exact)
                    const float k_fi = (float)(r & 0xFFFFU), k_th = (float)(r >> 16);
                    // b = Normalize(Cross(dir, dir + 0.01)) = Normalize(Cross(dir, (1,1,1))): the 0.01 and dir x dir cancel;
                    // c = Cross(dir, b) is already unit (dir, b orthonormal), so the reference's second Normalize is a no-op
                    f3 b = F3(axis.y - axis.z, axis.z - axis.x, axis.x - axis.y);
                    float bl2 = dot(b, b);
                    bool degenerate = !(bl2 > 0.0f);
                    // Normalize(0) = 0: the frame collapses for dir parallel to (1,1,1).  b is then exactly 0, so b * rsq(0 + 1e-30) = 0 too;
                    // and a non-zero bl2 (>= ~1e-15: squares of differences of unit-vector components) is unchanged by adding 1e-30
                    b = b * rsq_fast(bl2 + 1e-30f);
                    f3 c = cross(axis, b);
                    // fi = U*360/57.2958; cos from the hardware (input in revolutions), sin = +-sqrt(1 - cos^2) so that
                    // cos^2 + sin^2 = 1 to the last bit: v_sin/v_cos are each good to ~1e-5 only, and the sphere test
                    // amplifies an error in |L.d| by (distance/radius)^2
                    const float fi_rev = (float)((360.0 / 57.2958f) / 6.283185307179586);   // fi/(2*pi)
                    float x_rev = k_fi * (fi_rev * (1.0f / 65536.0f));
                    float cf = __builtin_amdgcn_cosf(x_rev);
                    float sf = sqrt_fast(fmaf(-cf, cf, 1.0f));
                    sf = __uint_as_float(__float_as_uint(sf) | (__float_as_uint(0.5f - x_rev) & 0x80000000U));   // negative beyond half a turn
                    float ct, st;
                    if (VARIANT == 3) {   // theta = U(-1,1)*thetaMax/57.2958
                        float th_rev = fmaf(k_th, 2.0f / 65536.0f, -1.0f) * m2.y * 0.15915494309189535f;
                        ct = __builtin_amdgcn_cosf(th_rev);
                        st = sqrt_fast(fmaf(-ct, ct, 1.0f));
                        st = __uint_as_float(__float_as_uint(st) | (__float_as_uint(th_rev) & 0x80000000U));      // |theta| <= 90 deg: sin has theta's sign
                    } else {              // theta = acos(U(cos(thetaMax/57.2958), 1))
                        ct = fmaf(m2.x, k_th, m2.y);      // m2.x = (1 - cone)/65536 (host-precomputed; exact power-of-two scaling)
                        st = sqrt_fast(fmaf(-ct, ct, 1.0f));   // ct <= 1 exactly (u_th < 1), so the fused 1 - ct*ct is never negative
                    }
                    nd = madd(madd(c, sf, b * cf), st, axis * ct);
                    // Ray's constructor normalises (Common/Ray.hpp:33-34); nd is unit by construction unless the frame
                    // collapsed, where it is axis*cos(theta)
                    if (degenerate) nd = normalize(nd);
                }
            }
            L.d = (REF || RT_NORM_DIRS(A)) ? strict_dir(nd) : nd;   // unit (REF: as Ray's constructor leaves it, Common/Ray.hpp:33-34): n is unit (slot layout), Reflect/Refract keep the length, the sampled direction is unit by construction
            if (mtype != MAT_DIELECTRIC) {
                f3 att = F3(m0);
                if (VARIANT == 4) att = att * fmaxf(dot(L.d, n), 0.0f);      // 04 Diffuse.hpp:77-78, Metal.hpp:83-84
                L.thr = L.thr * att;
            }
            ++L.depth;
        } else {
            if (VARIANT == 4 && mtype == MAT_LIGHT && !back) add_radiance<DEPOSIT>(L, L.thr * F3(m0));   // Light.hpp:47-54, Main.cpp:480-483
            L.alive = false;
        }
    }
}

// Small scenes (blob <= 48 KB, LDS = true): every block stages the whole blob in LDS once.  LDS = false: the blob is
// read from global memory with scalar loads (debug flag 2; A/B reference for the other two kernels).
// EARLY: wave-level early-outs in the primitive tests (rt_device.h), for scenes of many small primitives.
template <int VARIANT, bool LDS, bool EARLY = false, bool REF = RT_REF_RAYS>
__device__ __forceinline__ void path_body(const DevScene& SD, const PathArgs& A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
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
        regenerate<REF>(A, px, py, pkey, want, m_want, m_alive, L);
        if (L.alive) {
            // TraceTay: one scene query (03 Main.cpp:392, 04 Main.cpp:447)
            ++L.rays;
            float t;
#ifdef RT_JIT_SCENE   // scene-specialised build (rt_jit.hip): the scene query is generated straight-line code
            const uint32_t k = closest_hit_jit(L.o, L.d, A.t_min, A.t_min_pos, A.t_max, t);   // slot offset + SLOT_BYTES, 0 = miss
            const uint32_t hit = k + (slots0 - SLOT_BYTES);
            const bool miss = k == 0U;
            (void)runs;
#else
            uint32_t hit = closest_hit_slots<LDS, EARLY, REF>(M, runs, SD.n_sruns, slots0, L.o, L.d, A.t_min, A.t_min_pos, A.t_max, t);
            const bool miss = hit == 0U;
#endif
            shade<VARIANT, LDS, false, REF>(M, A, mats0, miss, hit, t, L);
        }
    }
    if (inside) {
        // debug flag 4: report the lane's scene-query count in w instead of its sample count (parity diagnostics)
        out[(size_t)slice * A.width * A.height + pixel] = make_float4(L.acc.x, L.acc.y, L.acc.z, (A.debug_flags & 4) ? (float)L.rays : (float)L.n_started);
    }
    wave_add_u64(counters + 2 * ((blockIdx.x + blockIdx.y) % RT_COUNTER_BANKS), L.rays);
}

// =================================================================================================
// Item form of the bounce loop (K3/K4): persistent waves, a work counter, lanes decoupled from pixels.
//
// path_body ties a lane to a pixel: a lane whose pixel is exhausted idles until the slowest pixel of its wave is done, a block
// lives as long as its slowest wave, long frames need sample slices (partial images in HBM plus a summing pass) to keep the
// launch tail short.  Here the grid is PERSISTENT (as many blocks as the chip holds) and every WAVE pulls units of work from one
// global counter until none is left.  A unit = a group of 2^unit_log2 consecutive pixels x the samples of one slice - normally
// the whole sample range, because a long frame is cut into units by making the pixel group SMALL (8 pixels at 1024 spp) instead
// of cutting the samples: a unit is then some 8-16 thousand paths whatever the frame, the tail of the launch is one unit long,
// and each pixel still has exactly one owner, so the sums are stored once, plainly - no partial images, no summing pass, HBM
// traffic = the algorithmic 16 B per pixel.  Inside a unit the wave's 64 lanes share its (pixel, sample) ITEMS: an idle lane
// takes the next item whichever pixel it belongs to (rank among the idle lanes by mbcnt, one scalar counter per wave), so no
// lane waits for another lane's pixel.  A finished path that carries radiance adds it to its pixel's sum in LDS as 32.32 fixed
// point (ds_add_u64): integer addition is order-independent, so the frame does not depend on which lane traced what - bit for
// bit the same frame from run to run and for every unit size (contributions are taken as >= 0 and < 2^32, resolution 2^-32;
// the host selects path_body for scenes with colours outside that).  No barrier after the scene blob is staged: waves are
// independent.  Same RNG keys (pixel, sample), same camera arithmetic, same query and shading code as path_body: the same
// paths; only the rounding of the per-pixel sum differs (exact fixed point rounded once vs a running float).
// =================================================================================================
#define RT_ITEMS_WAVE_BYTES (64 * 3 * 8 + 64 * 4 + 64 * 8)      /* per wave: fixed-point sums of up to 64 pixels, per-pixel query counts (debug flag 4), pixel table */
template <int VARIANT, bool LDS, bool EARLY = false>
__device__ __forceinline__ void path_items_body(const DevScene& SD, const PathArgs& A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    extern __shared__ float4 smem[];
    const char* base = reinterpret_cast<const char*>(SD.blob);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t ws_off = 0U;
    if (LDS) {
        for (int i = tid; i < SD.total_f4; i += 256) smem[i] = SD.blob[i];
        __syncthreads();                       // the only barrier: from here on the block's waves run independently
        base = reinterpret_cast<const char*>(smem);
        ws_off = (uint32_t)SD.total_f4 * 16U;
    }
    char* const ws = reinterpret_cast<char*>(smem) + ws_off + wave * RT_ITEMS_WAVE_BYTES;
    unsigned long long* const acc = reinterpret_cast<unsigned long long*>(ws);      // [pixel of the group][rgb]
    uint32_t* const qcount = reinterpret_cast<uint32_t*>(ws + 64 * 3 * 8);
    uint2* const pixtab = reinterpret_cast<uint2*>(ws + 64 * 3 * 8 + 64 * 4);      // per pixel of the unit: {RNG pixel key, x | y << 16 or ~0 = outside the image}
    const int4* __restrict__ runs = reinterpret_cast<const int4*>(SD.blob + SD.off_sruns);
    const SceneMem<LDS> M{base};
    const uint32_t slots0 = M.at((uint32_t)SD.off_slots * 16U), mats0 = M.at(0U);
    const int k = A.unit_log2, P = 1 << k;
    const uint32_t n_pixels = (uint32_t)A.width * (uint32_t)A.height;
    unsigned int rays = 0;
    unsigned int stat = 0;      // experiments (RT_MI355_JIT_DEFS=RT_STAT_ITER | RT_STAT_PASS): a wave-level event count, reported in the shadow-ray counter (tools/pool_stats.py)
    (void)stat;

    for (;;) {
        // ---- next unit of work: one atomic per wave and unit
        uint32_t u = 0U;
        if (lane == 0) u = (uint32_t)atomicAdd(counters + RT_WORK_COUNTER, 1ULL);
        u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
        if (u >= (uint32_t)A.n_units) break;
        uint32_t slice = 0U, group = u;
        if (A.n_units != A.n_groups) { slice = u / (uint32_t)A.n_groups; group = u - slice * (uint32_t)A.n_groups; }
        if (A.unit_order == 1) group = (uint32_t)A.n_groups - 1U - group;      // bottom-up: the launch ends with the cheap units of a frame that has sky on top (rt_api.hip)
        const int s_first = A.sample_begin + (int)slice * A.slice_len;
        const int s_end = min(s_first + A.slice_len, A.sample_end);
        const uint32_t first_px = group << k;                                  // the group's pixels: first_px .. first_px + P - 1, row-major
        const uint32_t by = first_px / (uint32_t)A.width, bx = first_px - by * (uint32_t)A.width;      // wave-uniform
        const uint32_t n_items = (uint32_t)max(s_end - s_first, 0) << k;
        uint32_t my_pixel = 0xFFFFFFFFU;       // lane i < P looks after pixel i of the group: table entry, zeroed sums, the store at the end
        if (lane < P) {
            uint32_t x = bx + (uint32_t)lane, y = by;
            while (x >= (uint32_t)A.width) { x -= (uint32_t)A.width; ++y; }
            const uint32_t pixel = y * (uint32_t)A.width + x;
            const bool inside = pixel < n_pixels;
            if (inside) my_pixel = pixel;
            pixtab[lane] = make_uint2(rng_pixel_key(A.seed, pixel), inside ? (x | (y << 16)) : 0xFFFFFFFFU);
            acc[3 * lane] = 0ULL; acc[3 * lane + 1] = 0ULL; acc[3 * lane + 2] = 0ULL; qcount[lane] = 0U;
        }
        __builtin_amdgcn_wave_barrier();       // this wave's LDS operations complete in order; the compiler must keep them so
        uint32_t next = 0U;                    // wave-uniform: the next item to hand out
        PathLane L = lane_init(0);
        L.acc0 = (uint32_t)(uintptr_t)(RT_LDS_AS char*)ws;

        for (;;) {
            const unsigned long long m_alive = __builtin_amdgcn_ballot_w64(L.alive);
            const bool left = next < n_items;
            if (m_alive == 0ULL && !left) break;
            const unsigned long long m_idle = ~m_alive;
#ifdef RT_STAT_ITER
            ++stat;
#endif
            if (left && (m_alive == 0ULL || __popcll(m_idle) >= A.regen_threshold)) {
#ifdef RT_STAT_PASS
                ++stat;
#endif
                // idle lanes take the next items in lane order: (pixel, sample) = (item mod P, item div P)
                const uint32_t item = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_idle, 0U));
                next += (uint32_t)__popcll(m_idle);
                if (!L.alive && item < n_items) {
                    L.pl = item & (uint32_t)(P - 1);
                    const uint2 pt = pixtab[L.pl];
                    if (pt.y != 0xFFFFFFFFU) {  // Render: one primary ray (03 Main.cpp:331-341, 04 Main.cpp:359-369); same arithmetic as regenerate()
                        const uint32_t x = pt.y & 0xFFFFU, y = pt.y >> 16;
                        L.skey = rng_sample_key(pt.x, (uint32_t)s_first + (item >> k));
                        L.rkey = L.skey + 5U * 0x9E3779B9U;
                        float bxj = 0.5f, byj = 0.5f;
                        if (A.jitter) rng_pair(L.skey, 0U, bxj, byj);
                        float cx = fmaf((float)(int)x + bxj, A.inv_w2, -1.0f) * A.kx;
                        float cy = -fmaf((float)(int)y + byj, A.inv_h2, -1.0f) * A.ky;
                        f3 v = F3(cx, cy, -1.0f);
                        const f3 rv = F3(dot(F3(A.r[0], A.r[1], A.r[2]), v), dot(F3(A.r[3], A.r[4], A.r[5]), v), dot(F3(A.r[6], A.r[7], A.r[8]), v));
                        L.d = rv * rsq_fast(dot(rv, rv));
                        L.o = F3(A.px, A.py, A.pz);
                        L.thr = F3(1, 1, 1);
                        L.depth = 0;
                        L.alive = true;
                    }
                }
            }
            if (L.alive) {
                // TraceTay: one scene query (03 Main.cpp:392, 04 Main.cpp:447)
                ++rays;
                if (A.debug_flags & 4) atomicAdd(&qcount[L.pl], 1U);
                float t;
#ifdef RT_JIT_SCENE
                const uint32_t kk = closest_hit_jit(L.o, L.d, A.t_min, A.t_min_pos, A.t_max, t);
                const uint32_t hit = kk + (slots0 - SLOT_BYTES);
                const bool miss = kk == 0U;
                (void)runs;
#else
                uint32_t hit = closest_hit_slots<LDS, EARLY>(M, runs, SD.n_sruns, slots0, L.o, L.d, A.t_min, A.t_min_pos, A.t_max, t);
                const bool miss = hit == 0U;
#endif
                shade<VARIANT, LDS, true>(M, A, mats0, miss, hit, t, L);      // a path that ends with radiance adds it to its pixel's sums in LDS
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (my_pixel != 0xFFFFFFFFU) {         // fixed point -> float, once per pixel
            const unsigned long long r = acc[3 * lane], g = acc[3 * lane + 1], b = acc[3 * lane + 2];
            auto tof = [](unsigned long long v) { return (float)((double)v * (1.0 / 4294967296.0)); };
            out[(size_t)slice * n_pixels + my_pixel] = make_float4(tof(r), tof(g), tof(b), (A.debug_flags & 4) ? (float)qcount[lane] : (float)max(s_end - s_first, 0));
        }
        __builtin_amdgcn_wave_barrier();
    }
    wave_add_u64(counters + 2 * ((blockIdx.x * 4 + wave) % RT_COUNTER_BANKS), rays);
#if defined(RT_STAT_ITER) || defined(RT_STAT_PASS)
    if (lane == 0) atomicAdd(counters + 2 * ((blockIdx.x * 4 + wave) % RT_COUNTER_BANKS) + 1, (unsigned long long)stat);
#endif
}

// =================================================================================================
// Queue form of the bounce loop (K3/K4): a wavefront pipeline inside the workgroup, with ray-queue
// compaction through LDS.
//
// path_body keeps a lane on its pixel: a lane whose path has ended idles until enough of its wave is
// idle to pay for a regeneration pass, and that pass then runs with most of the wave masked off.
// Here a workgroup (4 waves) owns a 16x16 pixel tile and a QUEUE of work items (pixel, sample) of it.
// Every iteration the workgroup compacts its live paths: each wave ballots its live lanes, takes a base
// from an LDS counter (one atomic per wave) and its live lanes write their 48-byte path state to slot
// base + mbcnt(ballot) of an LDS exchange buffer; after the barrier thread i < n_live reads state i and
// the threads behind them take the next 256 - n_live work items.  So every wave but (at most) one is
// full when it runs the scene query and the shading code, the camera code runs in whole waves, and no
// path state ever leaves the CU (12 KB of LDS per workgroup; HBM sees only the final sums).
// Paths are no longer tied to lanes, so a finished path hands its radiance over through the same
// mechanism: it appends (rgb, pixel) to a "finished" list (second counter, same buffer, filled from the
// end), and the first n_finished threads add the entries to the tile's accumulators in LDS.  Those are
// 32.32 FIXED-POINT sums (ds_add_u64): integer addition is order-independent, so the image does not
// depend on which wave got which base - reproducible bit for bit, run to run, like the rest.
// (Radiance contributions are therefore taken as >= 0 and < 2^32; resolution 2^-32.)
// Same RNG keys (pixel, sample), same camera arithmetic, same query and shading code as path_body: the
// paths are identical, only the rounding of the per-pixel sum differs (fixed point vs running float).
// =================================================================================================
#define RT_Q_XCH_BYTES (256 * 48)
#define RT_Q_WS_BYTES (RT_Q_XCH_BYTES + 256 * 3 * 8 + 256 * 8 + 64)      /* exchange + accumulators + pixel table + counters */
template <int VARIANT, bool LDS, bool EARLY = false>
__device__ __forceinline__ void path_queue_body(const DevScene& SD, const PathArgs& A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    extern __shared__ float4 smem[];
    const char* base = reinterpret_cast<const char*>(SD.blob);
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t ws_off = 0U;                       // the workspace sits behind the staged blob
    if (LDS) {
        for (int i = tid; i < SD.total_f4; i += 256) smem[i] = SD.blob[i];
        base = reinterpret_cast<const char*>(smem);
        ws_off = (uint32_t)SD.total_f4 * 16U;
    }
    char* const ws = reinterpret_cast<char*>(smem) + ws_off;
    float4* const xch = reinterpret_cast<float4*>(ws);                                        // 256 x 3 float4: live states from the front ...
    float4* const fin = reinterpret_cast<float4*>(ws + RT_Q_XCH_BYTES) - 1;                   // ... finished (rgb, pixel) entries from the back: fin[-k]
    unsigned long long* const acc = reinterpret_cast<unsigned long long*>(ws + RT_Q_XCH_BYTES);   // 256 x 3 fixed-point sums
    uint2* const pixtab = reinterpret_cast<uint2*>(ws + RT_Q_XCH_BYTES + 256 * 3 * 8);        // per tile pixel: {RNG pixel key, pixel index or ~0}
    uint32_t* const cnt = reinterpret_cast<uint32_t*>(ws + RT_Q_XCH_BYTES + 256 * 3 * 8 + 256 * 8);   // [2 phases][live, finished]
    const int4* __restrict__ runs = reinterpret_cast<const int4*>(SD.blob + SD.off_sruns);
    const SceneMem<LDS> M{base};
    const uint32_t slots0 = M.at((uint32_t)SD.off_slots * 16U), mats0 = M.at(0U);

    const int tiles_x = (A.width + 15) >> 4;
    const int tx16 = (blockIdx.x % tiles_x) * 16, ty16 = (blockIdx.x / tiles_x) * 16;
    const int slice = blockIdx.y;
    const int s_first = A.sample_begin + slice * A.slice_len;
    const int s_end = min(s_first + A.slice_len, A.sample_end);
    const int n_items = 256 * max(s_end - s_first, 0);
    {   // this thread's own pixel of the tile: table entry and zeroed accumulators
        const int px = tx16 + (tid & 15), py = ty16 + (tid >> 4);
        const bool inside = px < A.width && py < A.height;
        const uint32_t pixel = (uint32_t)(py * A.width + px);
        pixtab[tid] = make_uint2(rng_pixel_key(A.seed, pixel), inside ? pixel : 0xFFFFFFFFU);
        acc[3 * tid] = 0ULL; acc[3 * tid + 1] = 0ULL; acc[3 * tid + 2] = 0ULL;
        if (tid < 4) cnt[tid] = 0U;
    }
    PathLane L = lane_init(0);
    uint32_t pl = 0U;                           // tile pixel (0..255) of the lane's current path
    int item_base = 0, phase = 0;
    bool finished = false;                      // the lane's path ended in the last iteration and L.acc holds its radiance
    for (;;) {
        __syncthreads();                        // the exchange buffer is free (everyone has read the previous round), the counters of this phase are zero
        // ---- compaction: live states to the front of the exchange buffer, finished paths' radiance to its back
        const unsigned long long m_live = __builtin_amdgcn_ballot_w64(L.alive), m_fin = __builtin_amdgcn_ballot_w64(finished);
        uint32_t b_live = 0U, b_fin = 0U;
        if (lane == 0) {
            b_live = atomicAdd(&cnt[2 * phase], (uint32_t)__popcll(m_live));
            if (m_fin) b_fin = atomicAdd(&cnt[2 * phase + 1], (uint32_t)__popcll(m_fin));
        }
        b_live = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_live); b_fin = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_fin);
        if (L.alive) {
            const uint32_t r = b_live + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_live, 0U));
            float4* dst = xch + 3U * r;
            dst[0] = make_float4(L.o.x, L.o.y, L.o.z, L.d.x);
            dst[1] = make_float4(L.d.y, L.d.z, L.thr.x, L.thr.y);
            dst[2] = make_float4(L.thr.z, __uint_as_float(L.rkey), __int_as_float(L.depth), __uint_as_float(pl));
        } else if (finished) {
            const uint32_t r = b_fin + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_fin >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_fin, 0U));
            fin[-(int)r] = make_float4(L.acc.x, L.acc.y, L.acc.z, __uint_as_float(pl));
        }
        __syncthreads();                        // all states and entries are in place, the counters are final
        const int n_live = (int)cnt[2 * phase], n_fin = (int)cnt[2 * phase + 1];
        if (tid < 2) cnt[2 * (phase ^ 1) + tid] = 0U;       // the other phase's counters: nobody touches them before the next barrier
        phase ^= 1;
        if (n_live == 0 && item_base >= n_items) {          // workgroup-uniform: nothing in flight, nothing left to start
            if (tid < n_fin) {                              // the last finished paths
                const float4 e = fin[-tid];
                const uint32_t q = __float_as_uint(e.w);
                atomicAdd(&acc[3 * q], ((unsigned long long)(uint32_t)e.x << 32) | (uint32_t)(__builtin_amdgcn_fractf(e.x) * 4294967296.0f));
                atomicAdd(&acc[3 * q + 1], ((unsigned long long)(uint32_t)e.y << 32) | (uint32_t)(__builtin_amdgcn_fractf(e.y) * 4294967296.0f));
                atomicAdd(&acc[3 * q + 2], ((unsigned long long)(uint32_t)e.z << 32) | (uint32_t)(__builtin_amdgcn_fractf(e.z) * 4294967296.0f));
            }
            break;
        }
        float4 fe = make_float4(0, 0, 0, 0);
        if (tid < n_fin) fe = fin[-tid];                    // read before anything else: the slot may be overwritten next round only
        // ---- every thread takes a path: the first n_live an existing one, the rest new work items (camera code in whole waves)
        if (tid < n_live) {
            const float4 a = xch[3 * tid], b = xch[3 * tid + 1], c = xch[3 * tid + 2];
            L.o = F3(a.x, a.y, a.z); L.d = F3(a.w, b.x, b.y); L.thr = F3(b.z, b.w, c.x);
            L.rkey = __float_as_uint(c.y); L.depth = __float_as_int(c.z); pl = __float_as_uint(c.w);
            L.alive = true;
        } else {
            const int item = item_base + (tid - n_live);
            L.alive = false;
            if (item < n_items) {
                pl = (uint32_t)item & 255U;
                const uint2 pt = pixtab[pl];
                if (pt.y != 0xFFFFFFFFU) {      // Render: one primary ray (03 Main.cpp:331-341, 04 Main.cpp:359-369); same arithmetic as regenerate()
                    const int px = tx16 + (int)(pl & 15U), py = ty16 + (int)(pl >> 4);
                    L.skey = rng_sample_key(pt.x, (uint32_t)(s_first + (item >> 8)));
                    L.rkey = L.skey + 5U * 0x9E3779B9U;
                    float bx = 0.5f, by = 0.5f;
                    if (A.jitter) rng_pair(L.skey, 0U, bx, by);
                    float x = fmaf((float)px + bx, A.inv_w2, -1.0f) * A.kx;
                    float y = -fmaf((float)py + by, A.inv_h2, -1.0f) * A.ky;
                    f3 v = F3(x, y, -1.0f);
                    const f3 rv = F3(dot(F3(A.r[0], A.r[1], A.r[2]), v), dot(F3(A.r[3], A.r[4], A.r[5]), v), dot(F3(A.r[6], A.r[7], A.r[8]), v));
                    L.d = rv * rsq_fast(dot(rv, rv));
                    L.o = F3(A.px, A.py, A.pz);
                    L.thr = F3(1, 1, 1);
                    L.depth = 0;
                    L.alive = true;
                }
            }
        }
        item_base += 256 - n_live;
        L.acc = F3(0, 0, 0);
        // ---- the finished list: its first n_fin entries go to the tile's fixed-point sums (mostly one wave's work)
        if (tid < n_fin) {
            const uint32_t q = __float_as_uint(fe.w);
            atomicAdd(&acc[3 * q], ((unsigned long long)(uint32_t)fe.x << 32) | (uint32_t)(__builtin_amdgcn_fractf(fe.x) * 4294967296.0f));
            atomicAdd(&acc[3 * q + 1], ((unsigned long long)(uint32_t)fe.y << 32) | (uint32_t)(__builtin_amdgcn_fractf(fe.y) * 4294967296.0f));
            atomicAdd(&acc[3 * q + 2], ((unsigned long long)(uint32_t)fe.z << 32) | (uint32_t)(__builtin_amdgcn_fractf(fe.z) * 4294967296.0f));
        }
        // ---- TraceTay: one scene query and one bounce for every live lane
        finished = false;
        if (L.alive) {
            ++L.rays;
            float t;
#ifdef RT_JIT_SCENE
            const uint32_t k = closest_hit_jit(L.o, L.d, A.t_min, A.t_min_pos, A.t_max, t);
            const uint32_t hit = k + (slots0 - SLOT_BYTES);
            const bool miss = k == 0U;
            (void)runs;
#else
            uint32_t hit = closest_hit_slots<LDS, EARLY>(M, runs, SD.n_sruns, slots0, L.o, L.d, A.t_min, A.t_min_pos, A.t_max, t);
            const bool miss = hit == 0U;
#endif
            shade<VARIANT, LDS>(M, A, mats0, miss, hit, t, L);
            finished = !L.alive && (L.acc.x != 0.0f || L.acc.y != 0.0f || L.acc.z != 0.0f);     // black paths have nothing to hand over
        }
    }
    __syncthreads();
    {   // thread i writes tile pixel i: fixed point -> float (sum rgb, samples of this slice)
        const uint2 pt = pixtab[tid];
        if (pt.y != 0xFFFFFFFFU) {
            const unsigned long long r = acc[3 * tid], g = acc[3 * tid + 1], b = acc[3 * tid + 2];
            auto tof = [](unsigned long long v) { return (float)((double)v * (1.0 / 4294967296.0)); };
            out[(size_t)slice * A.width * A.height + pt.y] = make_float4(tof(r), tof(g), tof(b), (float)(n_items >> 8));
        }
    }
    wave_add_u64(counters + 2 * ((blockIdx.x + blockIdx.y) % RT_COUNTER_BANKS), L.rays);
}

// =================================================================================================
// Outcome-sorted form of the bounce loop (K3/K4): the workgroup sorts its 256 paths by what the scene
// query found, so that every wave runs ONE kind of work.
//
// On 03 a ray leaves to the sky 46 % of the time: in path_body / path_items_body each wave then runs the shading code with
// ~54 % of its lanes and the camera code with ~46 % (measured lane utilisation 0.70).  Here, after the query,
//   * a path that ENDS (miss: the sky term; a light; no material; the depth limit) hands its radiance to the pixel's fixed-point
//     sum in LDS at once and frees its lane - nothing of it is exchanged;
//   * a path that SCATTERS writes its 48-byte state (origin, direction, throughput, RNG key, depth | pixel | hit slot, t) to an
//     LDS exchange buffer, diffuse / metal hits from the front (rank = wave base from one LDS atomic + mbcnt of the ballot),
//     dielectric hits from the back;
// and after ONE barrier (the buffer and its counters are double / triple buffered) thread i < nA reads state i and shades it,
// the last nB threads shade the dielectric hits, and every thread in between starts a new path from the unit's next items - so
// shading, refraction and camera code each run in full waves (one boundary wave per class apart).  The workgroup pulls units
// of work (pixel group x sample range) from the global work counter like path_items_body's waves do; sums are the same 32.32
// fixed point, so the frame is bit-identical to the item form's (same paths, order-independent sums).
// =================================================================================================
template <int VARIANT, bool LDS, bool EARLY = false>
__device__ __forceinline__ void path_sorted_body(const DevScene& SD, const PathArgs& A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    extern __shared__ float4 smem[];
    const char* base = reinterpret_cast<const char*>(SD.blob);
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t ws_off = 0U;
    if (LDS) {
        for (int i = tid; i < SD.total_f4; i += 256) smem[i] = SD.blob[i];
        base = reinterpret_cast<const char*>(smem);
        ws_off = (uint32_t)SD.total_f4 * 16U;
    }
    char* const ws = reinterpret_cast<char*>(smem) + ws_off;
    float4* const xch = reinterpret_cast<float4*>(ws);                                                   // [2][256][3]
    unsigned long long* const acc = reinterpret_cast<unsigned long long*>(ws + 2 * RT_SORT_XCH_BYTES);   // [pixel][rgb]
    uint2* const pixtab = reinterpret_cast<uint2*>(ws + 2 * RT_SORT_XCH_BYTES + (1 << RT_SORT_MAX_LOG2) * 24);
    uint32_t* const cnt = reinterpret_cast<uint32_t*>(ws + 2 * RT_SORT_XCH_BYTES + (1 << RT_SORT_MAX_LOG2) * 32);   // [3 sets][2] + [6] unit word
    const int4* __restrict__ runs = reinterpret_cast<const int4*>(SD.blob + SD.off_sruns);
    const SceneMem<LDS> M{base};
    const uint32_t slots0 = M.at((uint32_t)SD.off_slots * 16U), mats0 = M.at(0U);
    const int k = A.unit_log2, P = 1 << k;
    const uint32_t n_pixels = (uint32_t)A.width * (uint32_t)A.height;
    unsigned int rays = 0;

    for (;;) {
        __syncthreads();                       // the previous unit's sums are stored (and, first time round, the blob is staged)
        if (tid == 0) cnt[6] = (uint32_t)atomicAdd(counters + RT_WORK_COUNTER, 1ULL);
        if (tid < 6) cnt[tid] = 0U;
        __syncthreads();
        const uint32_t u = cnt[6];
        if (u >= (uint32_t)A.n_units) break;
        uint32_t slice = 0U, group = u;
        if (A.n_units != A.n_groups) { slice = u / (uint32_t)A.n_groups; group = u - slice * (uint32_t)A.n_groups; }
        const int s_first = A.sample_begin + (int)slice * A.slice_len;
        const int s_end = min(s_first + A.slice_len, A.sample_end);
        const uint32_t first_px = group << k;
        const uint32_t by = first_px / (uint32_t)A.width, bx = first_px - by * (uint32_t)A.width;
        const uint32_t n_items = (uint32_t)max(s_end - s_first, 0) << k;
        uint32_t my_pixel = 0xFFFFFFFFU;
        if (tid < P) {
            uint32_t x = bx + (uint32_t)tid, y = by;
            while (x >= (uint32_t)A.width) { x -= (uint32_t)A.width; ++y; }
            const uint32_t pixel = y * (uint32_t)A.width + x;
            const bool inside = pixel < n_pixels;
            if (inside) my_pixel = pixel;
            pixtab[tid] = make_uint2(rng_pixel_key(A.seed, pixel), inside ? (x | (y << 16)) : 0xFFFFFFFFU);
            acc[3 * tid] = 0ULL; acc[3 * tid + 1] = 0ULL; acc[3 * tid + 2] = 0ULL;
        }
        PathLane L = lane_init(0);
        L.acc0 = (uint32_t)(uintptr_t)(RT_LDS_AS char*)acc;
        uint32_t item_base = 0U;
        int set = 0, buf = 0;
        uint32_t hit = 0U; float t = 0.0f;
        // (the first barrier of the bounce loop also publishes the table and the zeroed sums)

        for (;;) {
            // ---- scene query of every live path; paths that end are settled here, scattering paths are written out sorted
            bool hit_a = false, hit_b = false;
            if (L.alive) {
                ++rays;
#ifdef RT_JIT_SCENE
                const uint32_t kk = closest_hit_jit(L.o, L.d, A.t_min, A.t_min_pos, A.t_max, t);
                hit = kk + (slots0 - SLOT_BYTES);
                const bool miss = kk == 0U;
                (void)runs;
#else
                hit = closest_hit_slots<LDS, EARLY>(M, runs, SD.n_sruns, slots0, L.o, L.d, A.t_min, A.t_min_pos, A.t_max, t);
                const bool miss = hit == 0U;
#endif
                if (miss) {
                    if (VARIANT == 3) {   // sky (03 Main.cpp:430-432), as in shade()
                        float h = 0.5f * (L.d.y + 1.0f);
                        add_radiance<true>(L, L.thr * F3(fmaf(-0.5f, h, 1.0f), fmaf(-0.3f, h, 1.0f), 1.0f));
                    }
                } else {
                    const int mtype = __float_as_int(M.ld1(hit + SLOT_C + 12U));      // the slot's material class (compile_scene)
                    const bool scatters = (mtype == MAT_DIFFUSE) | (mtype == MAT_METAL) | (mtype == MAT_DIELECTRIC);
                    if (scatters & (L.depth < A.max_depth)) { hit_a = mtype != MAT_DIELECTRIC; hit_b = !hit_a; }
                    else if (VARIANT == 4 && mtype == MAT_LIGHT) {   // emits from its front face (Light.hpp:47-54) at any depth and ends the path; the hit record as in shade(), bit for bit
                        const float4 sa = M.ld4(hit + SLOT_A), sb = M.ld4(hit + SLOT_B), sc = M.ld4(hit + SLOT_C);
                        const f3 pc = madd(L.d, t, L.o) - F3(sa);
                        const float kq = sa.w * rsq_fast(fmaf(pc.z, pc.z, fmaf(pc.y, pc.y, fmaf(pc.x, pc.x, 1e-30f))));
                        if (!(dot(L.d, madd(pc, kq, F3(sc))) > 0.0f)) add_radiance<true>(L, L.thr * F3(M.ld4(mats0 + __float_as_uint(sb.w))));
                    }
                    // else: no material, or a bounce whose continuation the depth limit would cut (03 Main.cpp:383-386, 04 Main.cpp:438-441): the path ends black
                }
                L.alive = false;
            }
            const unsigned long long m_a = __builtin_amdgcn_ballot_w64(hit_a), m_b = __builtin_amdgcn_ballot_w64(hit_b);
            uint32_t b_a = 0U, b_b = 0U;
            if (lane == 0) {
                if (m_a) b_a = atomicAdd(&cnt[2 * set], (uint32_t)__popcll(m_a));
                if (m_b) b_b = atomicAdd(&cnt[2 * set + 1], (uint32_t)__popcll(m_b));
            }
            b_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_a); b_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_b);
            if (hit_a | hit_b) {
                const unsigned long long m = hit_a ? m_a : m_b;
                const uint32_t r = (hit_a ? b_a : b_b) + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0U));
                float4* dst = xch + (size_t)buf * 768 + 3U * (hit_a ? r : 255U - r);
                dst[0] = make_float4(L.o.x, L.o.y, L.o.z, L.d.x);
                dst[1] = make_float4(L.d.y, L.d.z, L.thr.x, L.thr.y);
                dst[2] = make_float4(L.thr.z, __uint_as_float(L.rkey), __uint_as_float((uint32_t)L.depth | (L.pl << 8) | ((hit - slots0) << 10)), t);   // slot offsets are multiples of 64: 6 spare bits
            }
            __syncthreads();                   // states and counters of this bounce are complete; the other buffer / the third counter set are free
            const uint32_t n_a = cnt[2 * set], n_b = cnt[2 * set + 1];
            { const int s2 = set == 0 ? 2 : set - 1; if (tid < 2) cnt[2 * s2 + tid] = 0U; }      // the set of the bounce before this one: everybody read it before this barrier
            if (n_a + n_b == 0U && item_base >= n_items) break;      // workgroup-uniform: nothing in flight, nothing left to start
            const bool take_a = (uint32_t)tid < n_a, take_b = (uint32_t)tid >= 256U - n_b;
            if (take_a | take_b) {
                const float4* src = xch + (size_t)buf * 768 + 3U * (uint32_t)tid;
                const float4 a = src[0], b = src[1], c = src[2];
                L.o = F3(a.x, a.y, a.z); L.d = F3(a.w, b.x, b.y); L.thr = F3(b.z, b.w, c.x);
                L.rkey = __float_as_uint(c.y);
                const uint32_t meta = __float_as_uint(c.z);
                L.depth = (int)(meta & 0xFFU); L.pl = (meta >> 8) & 0x7FU; hit = (meta >> 16 << 6) + slots0;
                L.alive = true;
                shade<VARIANT, LDS, true>(M, A, mats0, false, hit, c.w, L);
            } else {
                const uint32_t item = item_base + ((uint32_t)tid - n_a);
                if (item < n_items) {
                    L.pl = item & (uint32_t)(P - 1);
                    const uint2 pt = pixtab[L.pl];
                    if (pt.y != 0xFFFFFFFFU) {  // Render: one primary ray (03 Main.cpp:331-341, 04 Main.cpp:359-369); same arithmetic as regenerate()
                        const uint32_t x = pt.y & 0xFFFFU, y = pt.y >> 16;
                        L.skey = rng_sample_key(pt.x, (uint32_t)s_first + (item >> k));
                        L.rkey = L.skey + 5U * 0x9E3779B9U;
                        float bxj = 0.5f, byj = 0.5f;
                        if (A.jitter) rng_pair(L.skey, 0U, bxj, byj);
                        float cx = fmaf((float)(int)x + bxj, A.inv_w2, -1.0f) * A.kx;
                        float cy = -fmaf((float)(int)y + byj, A.inv_h2, -1.0f) * A.ky;
                        f3 v = F3(cx, cy, -1.0f);
                        const f3 rv = F3(dot(F3(A.r[0], A.r[1], A.r[2]), v), dot(F3(A.r[3], A.r[4], A.r[5]), v), dot(F3(A.r[6], A.r[7], A.r[8]), v));
                        L.d = rv * rsq_fast(dot(rv, rv));
                        L.o = F3(A.px, A.py, A.pz);
                        L.thr = F3(1, 1, 1);
                        L.depth = 0;
                        L.alive = true;
                    }
                }
            }
            item_base += 256U - (n_a + n_b);
            set = set == 2 ? 0 : set + 1; buf ^= 1;
        }
        __syncthreads();                       // every deposit of the unit has landed
        if (my_pixel != 0xFFFFFFFFU) {
            const unsigned long long r = acc[3 * tid], g = acc[3 * tid + 1], b = acc[3 * tid + 2];
            auto tof = [](unsigned long long v) { return (float)((double)v * (1.0 / 4294967296.0)); };
            out[(size_t)slice * n_pixels + my_pixel] = make_float4(tof(r), tof(g), tof(b), (float)max(s_end - s_first, 0));
        }
    }
    wave_add_u64(counters + 2 * ((blockIdx.x * 4 + (tid >> 6)) % RT_COUNTER_BANKS), rays);
}

// Large scenes (blob > 48 KB): the slot section is walked in pieces of <= RT_TILE_BYTES that the block's four waves
// stage through LDS together (one global read per block instead of one per wave, and VGPR operands for the tests), so
// the bounce loop runs in lockstep per BLOCK: regenerate per wave, then every chunk between two barriers, then shade.
// The winner is remembered as a blob offset (its chunk is gone from LDS by shading time) and shaded from global memory.
template <int VARIANT, bool REF = RT_REF_RAYS>
__device__ __forceinline__ void path_tiled_body(const DevScene& SD, const PathArgs& A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    extern __shared__ float4 smem[];
    const int4* __restrict__ runs = reinterpret_cast<const int4*>(SD.blob + SD.off_sruns);
    const int4* __restrict__ chunks = reinterpret_cast<const int4*>(SD.blob + SD.off_chunks);
    const SceneMem<false> G{reinterpret_cast<const char*>(SD.blob)};
    const SceneMem<true> T{reinterpret_cast<const char*>(smem)};
    // chunks are staged one slot past the start of LDS: closest_hit_runs reports the winner as its LDS address and 0 means
    // "nothing hit", so no slot may sit at address 0 (the launch reserves RT_TILE_BYTES + SLOT_BYTES)
    const uint32_t tile0 = T.at((uint32_t)SLOT_BYTES);
    float4* const tile = smem + SLOT_BYTES / 16;
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
            regenerate<REF>(A, px, py, pkey, want, m_want, m_alive, L);
        }
        if (!__syncthreads_or(L.alive)) break;      // block-uniform: a finished wave keeps staging chunks for the others
        float closest = A.t_max;
        uint32_t best = 0U;
        const f3 inv_d = F3(rcp_fast(L.d.x), rcp_fast(L.d.y), rcp_fast(L.d.z));
        for (int c = 0; c < SD.n_chunks; ++c) {
            const int4 ch = chunks[c];
            const float4* __restrict__ src = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(SD.blob) + ch.x);
            if (c) __syncthreads();                 // everyone is done with the previous chunk
            for (int i = threadIdx.x; i < (ch.y >> 4); i += 256) tile[i] = src[i];
            __syncthreads();
            if (L.alive) {
                uint32_t b = 0U;
                closest_hit_runs<true, false, true, REF>(T, runs + ch.z, ch.w, tile0, L.o, L.d, inv_d, A.t_min, A.t_min_pos, closest, b);
                if (b) best = b - tile0 + (uint32_t)ch.x;
            }
        }
        if (L.alive) {
            ++L.rays;
            shade<VARIANT, false, false, REF>(G, A, 0U, best == 0U, best, closest, L);
        }
    }
    if (inside) out[(size_t)slice * A.width * A.height + pixel] = make_float4(L.acc.x, L.acc.y, L.acc.z, (A.debug_flags & 4) ? (float)L.rays : (float)L.n_started);
    wave_add_u64(counters + 2 * ((blockIdx.x + blockIdx.y) % RT_COUNTER_BANKS), L.rays);
}

} // namespace rt
