// rt_whitted_device.h - device code of the Whitted kernels (01/02).  Included by rt_kernels.hip and, as text, by the
// scene-specialised hiprtc build of the 02 kernel (rt_jit.hip), so it must compile without any system include.
#pragma once
#include "rt_args.h"
#include "rt_device.h"
#include "rt_strict.h"

namespace rt {

#define WBLOCK 128
// The specular term's powf (01 Main.cpp:441, 02 Main.cpp:472).  The reference's value comes from its platform's libm; glibc's powf - what the oracle and the
// reference-made goldens use - is correctly rounded in all but ~1e-4 of its results, the device library's is good to 1-2 ulp: that was the last bit in which 2.2 % of
// 01's pixels differed from the reference-made frame.  Shininess is a whole number in every scene of the reference (16, 128, 512): x^n by squaring in DOUBLE (<= 2 log2 n
// roundings of 1.1e-16) rounded to float once is the correctly rounded float but for ~1e-7 of the arguments, i.e. glibc's value.  Other exponents keep the device powf
// (an inlined double pow() costs the kernels 50-60 VGPRs: measured, 02 then spills 192 B per lane).
__device__ __forceinline__ float pow_specular(float x, float y) {
    if (y >= 0.0f && y <= 65536.0f && y == truncf(y)) {
        double b = (double)x, r = 1.0;
        for (uint32_t n = (uint32_t)y; n != 0U; n >>= 1) { if (n & 1U) r *= b; b *= b; }
        return (float)r;
    }
    return powf(x, y);
}
// WALK = false: the whole of TraceTay in one kernel (shadow rays traced where the hit is shaded).
// WALK = true: stage 1 of the WAVEFRONT form - the tree walk only.  Where TraceTay traces the shadow rays of a shaded hit
// (01 Main.cpp:435-436, 02 Main.cpp:452-467) this stage writes a request {hit point, pixel, unshadowed contribution of the light,
// tree node | light, direction and distance to the light} to a queue in HBM and goes on with the tree; stage 2
// (whitted_shadow_body) traces the requests' shadow rays with one thread per ray - the whole chip works on them even when the
// frame has few pixels - and counts the occluded ones per request; stage 3 (whitted_combine_body) adds every request's
// contribution, scaled by its visibility, to its pixel in the order the walk emitted them.  The walk needs none of the shadow
// query's registers (round 2: 168 B per lane of scratch in 02's kernel, 56 B in 01's, at the 64 VGPRs of 8 waves per SIMD).
template <bool SOFT, bool WALK = false>
__device__ __forceinline__ void whitted_body(const DevScene& SD, const WhittedArgs& A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    const Scn S = scn_over(SD, SD.blob);
    // slot k (k = depth of the parked child, 1..max_depth+1): o(3) d(3) weight node, one column per thread; sized by the
    // launch to max_depth + 2 slots (4 KB each), so the default depth 4 leaves room for 6 blocks per CU
    extern __shared__ float stk_raw[];
    float (*stk)[8][WBLOCK] = reinterpret_cast<float (*)[8][WBLOCK]>(stk_raw);
    // 02: the shadow samples (almost all of its rays) run on the slot layout staged in LDS behind the stack, with the
    // fast tests of the path kernels (see fastshadow in rt_strict.h for why that is safe); A.shadow_lds = 0 when the
    // scene is too large for that
    const SceneMem<true> T{reinterpret_cast<const char*>(stk_raw)};
    // the staged slots start one slot past the stack so that no slot ever sits at LDS address 0: closest_hit_runs reports
    // the winner as its address and 0 means "nothing hit" (with no stack, stack_bytes = 0, the first slot would be at 0)
    const uint32_t sh_slots = T.at((uint32_t)A.stack_bytes + SLOT_BYTES);
    const int4* __restrict__ sruns = reinterpret_cast<const int4*>(SD.blob + SD.off_sruns);
#ifndef RT_JIT_SCENE
    if (SOFT && A.shadow_lds) {
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<char*>(stk_raw) + A.stack_bytes + SLOT_BYTES);
        const float4* src = SD.blob + SD.off_slots;
        for (int i = threadIdx.x; i < (SD.slot_bytes >> 4); i += WBLOCK) dst[i] = src[i];
        __syncthreads();
    }
#else   // scene-specialised build (rt_jit.hip): the shadow samples' scene query is generated code, nothing to stage
    (void)T; (void)sh_slots; (void)sruns;
#endif
    const int tid = threadIdx.x;
    // 16x8 pixels per 128-thread block, 8x8 per wave
    int tiles_x = (A.width + 15) >> 4;
    int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    int px = tx * 16 + (tid >> 6) * 8 + (tid & 7), py = ty * 8 + ((tid & 63) >> 3);
    bool inside = px < A.width && py < A.height;
    const bool in_frame = inside;
    const uint32_t pixel = (uint32_t)(py * A.width + px);
    const uint32_t skey = rng_sample_key(rng_pixel_key(A.seed, pixel), 0U);

    if (!WALK && A.only_marked) inside = inside && A.wf_head[pixel] == RT_WF_OVERFLOW;      // second pass: pixels whose requests did not fit the queue
    uint32_t wf_head = RT_WF_NONE;      // WALK: the pixel's latest request (a chain through wf_prev, newest first)
    bool wf_overflow = false;
    uint32_t base_used = 0U;            // WALK: which of the pixel's primary-hit slots (one per light) hold a request
    f3 o = F3(0, 0, 0), d = strict::primary_dir(px, py, A.w, A.h, A.tan_half);     // 01 Main.cpp:340-344
    float weight = 1.0f, tmin = 0.0f;
    int depth = 0;
    uint32_t node = 1U, pending = 0U;
    f3 acc = F3(0, 0, 0);
    unsigned int rays = 0, srays = 0;
    bool active = inside;
    const f3 sky = F3(0.2f, 0.7f, 0.8f);
    // Monolithic kernels (round 3): the colour is evaluated as the reference evaluates it - bottom-up, Mix(primary, Mix(refl, refr, wr), wp) per level in its own operation
    // order (01 Main.cpp:480-484) - instead of as a weighted sum of the nodes' contributions, which is the same number only in exact arithmetic (it was the last bit in
    // which 2.2 % of 01's pixels still differed from the reference-made frame).  Each level of the current root-to-node path leaves {primary, material} in LDS (4 floats);
    // when a node has returned, the lane folds the levels back from the deepest.  A chain (no material spawns reflection AND refraction at one hit - none of the
    // reference's does) needs nothing else: 10 KB per block at depth 4, 14 blocks = 7 waves per SIMD per CU.  A branching tree parks the refraction child of a node while
    // its reflection subtree is walked (8 floats per level behind the records); when the fold reaches such a level it leaves the reflection's colour in the parked
    // child's place, walks the refraction subtree, and folds on from there with both colours in hand: post-order, as the recursion returns.  The wavefront walk keeps
    // the weighted sum (its shadow terms are only known after stage 2).
    constexpr bool chain = !WALK;
    float (*lvl)[4][WBLOCK] = reinterpret_cast<float (*)[4][WBLOCK]>(stk_raw);
    // parked children: WALK - slot cd of the stack; else slot cd - 1 behind the max_depth + 1 level records
    float (*park)[8][WBLOCK] = WALK ? stk : reinterpret_cast<float (*)[8][WBLOCK]>(stk_raw + (A.max_depth + 1) * 4 * WBLOCK) - 1;
    uint32_t has_refl = 0U;             // bit lv: level lv's reflection colour is known (kept in the first three floats of park[lv + 1])

    while (__any(active)) {
        if (active) {
            bool done_node = true;   // pop after this node unless a child continues in place
            float t; int idx = -1;
#ifdef RT_JIT_STRICT_SCENE   // scene-specialised build: the same strict tests, one call per primitive with literal operands (rt_jit.hip)
            if (depth <= A.max_depth) { ++rays; idx = closest_hit_strict_jit(o, d, tmin, 1000.0f, t); }
#else
            if (depth <= A.max_depth) { ++rays; idx = strict::closest_hit(S, o, d, tmin, 1000.0f, t); }          // :386-403
#endif
            f3 leaf = F3(0, 0, 0);   // CHAIN mode: the colour of the node the chain ends in (a hit without a child leaves its own record and a black child)
            if (idx < 0) {
                if (chain) leaf = sky; else acc = madd(sky, weight, acc);                                // :489-493
            } else {
                int4 inf = S.info[idx];
                f3 p = strict::add(o, strict::muls(d, t));                                               // :414
                f3 n = (inf.x == PRIM_SPHERE) ? strict::normalize(strict::sub(p, F3(S.geom[inf.y]))) : F3(S.geom[inf.y]);   // Sphere.hpp:48, Plane.hpp:48
                float4 m0 = S.mats[inf.z * 3], m1 = S.mats[inf.z * 3 + 1], m2 = S.mats[inf.z * 3 + 2];
                float p2s = m1.z, r2r = m1.w;
                float wp = fmaxf(1.0f - p2s, 0.0f), wr = fmaxf(1.0f - r2r, 0.0f);
                f3 primary = F3(0, 0, 0);
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
                        if (WALK) {
                            if (depth + 1 <= A.max_depth) {      // shadow rays apply: hand the light's contribution to the shadow stage
                                float dd = fmaxf(0.0f, strict::dot(tod, n));
                                float sdot = fmaxf(0.0f, strict::dot(strict::reflect(-tod, n), -d));
                                float sp = pow_specular(sdot, m1.y);
                                const f3 cl = (F3(l1) * dd * F3(m0) + F3(l1) * sp * m1.x) * (weight * (1.0f - wp));
                                // slot: a primary hit's request sits at pixel * lights + light (no atomic: 8 million of them on one
                                // counter would cost more than the frame); deeper nodes draw from the region behind, one atomic per wave
                                uint32_t slot;
                                if (node == 1U && li < 32) { slot = pixel * (uint32_t)S.n_lights + (uint32_t)li; base_used |= 1U << li; }
                                else {
                                    const unsigned long long m = __builtin_amdgcn_ballot_w64(true);      // the lanes that are here together
                                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0U));
                                    uint32_t b0 = 0U;
                                    if (rank == 0U) b0 = atomicAdd(A.wf_counter, (uint32_t)__popcll(m));
                                    b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
                                    slot = b0 + rank < A.wf_capacity ? A.wf_base + b0 + rank : 0xFFFFFFFFU;
                                }
                                if (slot != 0xFFFFFFFFU) {
                                    A.wf_req[3 * (size_t)slot] = make_float4(p.x, p.y, p.z, __uint_as_float(pixel));
                                    A.wf_req[3 * (size_t)slot + 1] = make_float4(cl.x, cl.y, cl.z, __uint_as_float((node << 16) | (uint32_t)li));
                                    A.wf_req[3 * (size_t)slot + 2] = make_float4(tod.x, tod.y, tod.z, len);
                                    A.wf_prev[slot] = wf_head; A.wf_occ[slot] = 0U;
                                    wf_head = slot;
                                } else wf_overflow = true;
                                continue;
                            }
                        } else if (!SOFT) {                                                                     // 01 :435-436
#ifdef RT_JIT_STRICT_SCENE
                            if (depth + 1 <= A.max_depth) { ++srays; skip = any_hit_strict_jit(p, tod, 0.01f, len); }
#else
                            if (depth + 1 <= A.max_depth) { ++srays; skip = strict::any_hit(S, p, tod, 0.01f, len); }
#endif
                        } else if (depth + 1 <= A.max_depth) {                                           // 02 :452-467, :539-566
                            float per = 1.0f / (float)A.shadow_samples;
                            float tmax_s = len - l0.w;
                            const float radius_s = l0.w * (1.0f / 65536.0f);
                            for (int i = 0; i < A.shadow_samples; ++i) {
                                uint32_t c = ((node * (uint32_t)S.n_lights + (uint32_t)li) * (uint32_t)A.shadow_samples + (uint32_t)i) * 4U;
                                float k0, k1, k2, k3;
                                rng_pair_raw(skey, c, k0, k1); rng_pair_raw(skey, c + 2U, k2, k3);
                                f3 sd = fastshadow::to_light_disk_raw(p, lp, tod, radius_s, k0, k1, k2, k3);
                                ++srays;
                                bool occluded;
#ifdef RT_JIT_SCENE
                                occluded = any_hit_jit(p, sd, 0.01f, 0.01f, tmax_s);
#else
                                if (A.shadow_lds) {
                                    float closest = tmax_s; uint32_t best = 0U;
                                    closest_hit_runs<true, true>(T, sruns, SD.n_sruns, sh_slots, p, sd, F3(rcp_fast(sd.x), rcp_fast(sd.y), rcp_fast(sd.z)), 0.01f, 0.01f, closest, best, SD.srun_pack[0], SD.srun_pack[1]);
                                    occluded = best != 0U;
                                } else occluded = fastshadow::any_hit(S, p, sd, 0.01f, tmax_s);
#endif
                                if (occluded) intensity -= per;
                            }
                        }
                        if (skip) continue;   // 01: an occluded light is skipped entirely
                        float dd = fmaxf(0.0f, strict::dot(tod, n));
                        float sdot = fmaxf(0.0f, strict::dot(strict::reflect(-tod, n), -d));
                        float sp = pow_specular(sdot, m1.y);
                        if (SOFT) {                                                                      // 02 :474-477: (colour * term) * intensity
                            diffuse = strict::add(diffuse, strict::muls(strict::muls(F3(l1), dd), intensity));
                            specular = strict::add(specular, strict::muls(strict::muls(F3(l1), sp), intensity));
                        } else {                                                                         // 01 :439-442
                            diffuse = strict::add(diffuse, strict::muls(F3(l1), dd));
                            specular = strict::add(specular, strict::muls(F3(l1), sp));
                        }
                    }
                    primary = strict::add(strict::mulv(diffuse, F3(m0)), strict::muls(specular, m1.x));  // :446
                    if (!chain) acc = madd(primary, weight * (1.0f - wp), acc);
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
                        park[cd][0][tid] = p.x; park[cd][1][tid] = p.y; park[cd][2][tid] = p.z;
                        park[cd][3][tid] = d_refr.x; park[cd][4][tid] = d_refr.y; park[cd][5][tid] = d_refr.z;
                        park[cd][6][tid] = w_refr; park[cd][7][tid] = __uint_as_float(node * 2U + 1U);
                        pending |= 1U << cd;
                    }
                    if (do_refl || do_refr) {
                        o = p; tmin = 0.001f; depth = cd;
                        if (do_refl) { d = d_refl; weight = w_refl; node = node * 2U; }
                        else         { d = d_refr; weight = w_refr; node = node * 2U + 1U; }
                        done_node = false;
                    }
                }
                if (chain) {             // this level's record ('depth' has already moved on when a child continues); wp, wr and which child there is follow from the material's p2s, r2r
                    const int lv = done_node ? depth : depth - 1;
                    lvl[lv][0][tid] = primary.x; lvl[lv][1][tid] = primary.y; lvl[lv][2][tid] = primary.z;
                    lvl[lv][3][tid] = __int_as_float(inf.z);
                }
            }
            if (done_node && chain) {
                // a node has returned: a miss / beyond the depth limit (the sky; levels 0 .. depth-1 above it) or a hit without a child (its own level on top).
                // TraceTay's return path, level by level (01 Main.cpp:450-484): secondary = Mix(refl, refr, wr) with the children's colours (black for a child
                // that was not traced), colour = Mix(primary, secondary, wp); a level without a secondary term mixes with black at wp = 0, as the reference does.
                f3 c = leaf;
                bool resumed = false;
                for (int lv = (idx < 0) ? depth - 1 : depth; lv >= 0; --lv) {
                    if (pending & (2U << lv)) {      // c is the reflection's colour and the refraction child is still parked: keep c in its place and walk that subtree
                        pending &= ~(2U << lv); has_refl |= 1U << lv;
                        o = F3(park[lv + 1][0][tid], park[lv + 1][1][tid], park[lv + 1][2][tid]);
                        d = F3(park[lv + 1][3][tid], park[lv + 1][4][tid], park[lv + 1][5][tid]);
                        node = __float_as_uint(park[lv + 1][7][tid]);
                        park[lv + 1][0][tid] = c.x; park[lv + 1][1][tid] = c.y; park[lv + 1][2][tid] = c.z;
                        depth = lv + 1; tmin = 0.001f;
                        resumed = true;
                        break;
                    }
                    const f3 prim = F3(lvl[lv][0][tid], lvl[lv][1][tid], lvl[lv][2][tid]);
                    const float4 lm1 = S.mats[__float_as_int(lvl[lv][3][tid]) * 3 + 1];
                    const float lp2s = lm1.z, lr2r = lm1.w;
                    f3 secondary = F3(0, 0, 0);
                    if (lp2s < 1.0f) {
                        f3 refl = c, refr = F3(0, 0, 0);                   // the only child is the reflection
                        if (has_refl & (1U << lv)) {                       // both children: c is the refraction's colour
                            refl = F3(park[lv + 1][0][tid], park[lv + 1][1][tid], park[lv + 1][2][tid]); refr = c;
                            has_refl &= ~(1U << lv);
                        } else if (!(lr2r > 0.0f)) { refl = F3(0, 0, 0); refr = c; }      // the only child is the refraction
                        secondary = strict::mix(refl, refr, fmaxf(1.0f - lr2r, 0.0f));
                    }
                    c = strict::mix(prim, secondary, fmaxf(1.0f - lp2s, 0.0f));
                }
                if (!resumed) { acc = c; active = false; }
            } else if (done_node) {
                if (pending == 0U) active = false;
                else {
                    int k = 31 - __clz((int)pending);
                    pending &= ~(1U << k);
                    o = F3(park[k][0][tid], park[k][1][tid], park[k][2][tid]);
                    d = F3(park[k][3][tid], park[k][4][tid], park[k][5][tid]);
                    weight = park[k][6][tid]; node = __float_as_uint(park[k][7][tid]);
                    depth = k; tmin = 0.001f;
                }
            }
        }
    }
    if (WALK) {
        if (in_frame) {
            A.wf_head[pixel] = wf_overflow ? RT_WF_OVERFLOW : wf_head;
            for (int li = 0; li < S.n_lights; ++li)      // primary-hit slots without a request (sky, a material without a primary term, an overflowed pixel) are marked empty
                if (wf_overflow || !((base_used >> li) & 1U)) A.wf_req[3 * ((size_t)pixel * S.n_lights + li) + 1].w = __uint_as_float(0xFFFFFFFFU);
        }
        if (wf_overflow) rays = 0;      // the monolithic pass renders (and counts) this pixel from scratch
    }
    if (inside) out[pixel] = make_float4(acc.x, acc.y, acc.z, 1.0f);
    wave_add_u64(counters + 2 * (blockIdx.x % RT_COUNTER_BANKS), rays);
    wave_add_u64(counters + 2 * (blockIdx.x % RT_COUNTER_BANKS) + 1, srays);
}

// Stage 2 of the wavefront form: one thread per shadow ray (SOFT: request x sample, the samples of a request side by side;
// else one ray per request), a grid-stride loop over however many requests stage 1 emitted.  The same arithmetic as the
// monolithic kernel's shadow code: strict any-hit for 01 (its result gates the light), the fast tests for 02's samples.
template <bool SOFT>
__device__ __forceinline__ void whitted_shadow_body(const DevScene& SD, const WhittedArgs& A) {
    const Scn S = scn_over(SD, SD.blob);
    extern __shared__ float lds_raw[];
    const SceneMem<true> T{reinterpret_cast<const char*>(lds_raw)};
    const uint32_t sh_slots = T.at((uint32_t)SLOT_BYTES);      // no slot at LDS address 0 (0 = "nothing hit")
    const int4* __restrict__ sruns = reinterpret_cast<const int4*>(SD.blob + SD.off_sruns);
#ifndef RT_JIT_SCENE
    if (SOFT && A.shadow_lds) {
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<char*>(lds_raw) + SLOT_BYTES);
        const float4* src = SD.blob + SD.off_slots;
        for (int i = threadIdx.x; i < (SD.slot_bytes >> 4); i += 256) dst[i] = src[i];
        __syncthreads();
    }
#else
    (void)T; (void)sh_slots; (void)sruns;
#endif
    const uint32_t n_req = A.wf_base + min(*A.wf_counter, A.wf_capacity);      // the primary-hit slots (one per pixel and light, some empty) + the deeper nodes' requests
    // SOFT: G = 2^g threads share a request (g = A.wf_group_log2, G <= min(samples, 64)); thread j of them traces samples j, j + G, ...
    const int g = SOFT ? A.wf_group_log2 : 0;
    const uint32_t n_threads = n_req << g;      // < 2^32: the host checks
    for (uint32_t it = blockIdx.x * 256U + threadIdx.x; it < n_threads; it += gridDim.x * 256U) {
        const uint32_t slot = it >> g;
        const float4 ra = A.wf_req[3 * (size_t)slot], rb = A.wf_req[3 * (size_t)slot + 1], rc = A.wf_req[3 * (size_t)slot + 2];
        if (__float_as_uint(rb.w) == 0xFFFFFFFFU) continue;      // an empty primary-hit slot
        const f3 p = F3(ra), tod = F3(rc);
        const float len = rc.w;
        uint32_t n_occ = 0U;
        if (!SOFT) n_occ = strict::any_hit(S, p, tod, 0.01f, len) ? 1U : 0U;                            // 01 :435-436
        else {                                                                                           // 02 :452-467, :539-566
            const uint32_t pixel = __float_as_uint(ra.w), meta = __float_as_uint(rb.w);
            const uint32_t node = meta >> 16, li = meta & 0xFFFFU;
            const uint32_t skey = rng_sample_key(rng_pixel_key(A.seed, pixel), 0U);
            const float4 l0 = S.lights[2 * li];
            const float tmax_s = len - l0.w, radius_s = l0.w * (1.0f / 65536.0f);
            const uint32_t c0 = (node * (uint32_t)S.n_lights + li) * (uint32_t)A.shadow_samples;
            for (uint32_t i = it & ((1U << g) - 1U); i < (uint32_t)A.shadow_samples; i += 1U << g) {
                const uint32_t c = (c0 + i) * 4U;
                float k0, k1, k2, k3;
                rng_pair_raw(skey, c, k0, k1); rng_pair_raw(skey, c + 2U, k2, k3);
                const f3 sd = fastshadow::to_light_disk_raw(p, F3(l0), tod, radius_s, k0, k1, k2, k3);
                bool occluded;
#ifdef RT_JIT_SCENE
                occluded = any_hit_jit(p, sd, 0.01f, 0.01f, tmax_s);
#else
                if (A.shadow_lds) {
                    float closest = tmax_s; uint32_t best = 0U;
                    closest_hit_runs<true, true>(T, sruns, SD.n_sruns, sh_slots, p, sd, F3(rcp_fast(sd.x), rcp_fast(sd.y), rcp_fast(sd.z)), 0.01f, 0.01f, closest, best, SD.srun_pack[0], SD.srun_pack[1]);
                    occluded = best != 0U;
                } else occluded = fastshadow::any_hit(S, p, sd, 0.01f, tmax_s);
#endif
                n_occ += occluded ? 1U : 0U;
            }
        }
        if (n_occ) atomicAdd(&A.wf_occ[slot], n_occ);      // an integer count: order-independent
    }
}

// Stage 3: one thread per pixel adds its requests' contributions, each scaled by its visibility - 02: intensity = 1 - 1/S per
// occluded sample, subtracted one by one as the reference does (02 Main.cpp:465); 01: an occluded light is skipped (:437) -
// newest request first, a fixed order per pixel; and counts the shadow rays of the requests it owns.
template <bool SOFT>
__device__ __forceinline__ void whitted_combine_body(const WhittedArgs& A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    const uint32_t pixel = blockIdx.x * 256U + threadIdx.x;
    unsigned int srays = 0;
    if (pixel < (uint32_t)A.width * (uint32_t)A.height) {
        uint32_t h = A.wf_head[pixel];
        if (h != RT_WF_NONE && h != RT_WF_OVERFLOW) {
            float4 px = out[pixel];
            f3 acc = F3(px);
            const float per = 1.0f / (float)A.shadow_samples;
            while (h != RT_WF_NONE) {
                const float4 rb = A.wf_req[3 * (size_t)h + 1];
                const uint32_t occ = A.wf_occ[h];
                float intensity = 1.0f;
                if (SOFT) { for (uint32_t k = 0; k < occ; ++k) intensity -= per; srays += (unsigned int)A.shadow_samples; }
                else { intensity = occ ? 0.0f : 1.0f; ++srays; }
                acc = madd(F3(rb), intensity, acc);
                h = A.wf_prev[h];
            }
            out[pixel] = make_float4(acc.x, acc.y, acc.z, 1.0f);
        }
    }
    wave_add_u64(counters + 2 * (blockIdx.x % RT_COUNTER_BANKS) + 1, srays);
}

} // namespace rt
