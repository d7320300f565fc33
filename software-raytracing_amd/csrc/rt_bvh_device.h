// rt_bvh_device.h - the optional BVH scene query of the path kernels (rt_set_option(RT_OPT_ACCEL, 1)).
// It must return what the brute-force loop (closest_hit_runs, rt_device.h) returns, bit for bit:
//  * every primitive test below is the SAME expression tree as in closest_hit_runs (explicit fmaf, so the same bits);
//  * the brute-force loop walks the elements in scene order against a shrinking tMax and accepts t <= tMax, so among
//    equal distances the LAST element wins; a tree walk meets the elements in another order, so it keeps the winner by
//    the order-free rule "smaller t, or equal t and larger slot offset" (slot offsets grow in scene order);
//  * Ray::intersectsSphere reports a hit AT tMax for a ray that starts inside a sphere whose far root lies beyond tMax
//    (Common/Ray.hpp:127-134): the element takes the hit, the distance stays.  In scene order the last "taker" wins:
//    with (t*, p) the order-free winner, that is the largest slot S > p among the spheres that contain the origin and
//    have their far root beyond t* (rt_bvh.h: their boxes contain the origin, so the walk always reaches them).
#pragma once
#include "rt_path_device.h"

namespace rt {

struct BvhHit { float closest; uint32_t best; uint32_t steal; float steal_t; };

__device__ __forceinline__ void bvh_take(BvhHit& H, bool ok, float t, uint32_t q) {
    const bool hit = ok & ((t < H.closest) | ((t == H.closest) & (q > H.best)));
    H.closest = hit ? t : H.closest; H.best = hit ? q : H.best;
}

// One slot against the ray; ref = blob byte offset of the slot | device type.
__device__ __forceinline__ void bvh_test(const SceneMem<false> M, uint32_t ref, f3 o, f3 d, f3 inv_d, float tmin, float tminp, BvhHit& H) {
    const uint32_t q = ref & ~63U;
    const int type = (int)(ref & 15U);
    if (is_sphere_type(type)) {
        float4 g = M.ld4(q);
        float t1, t2;      // as in closest_hit_runs: same bits
        if (type == PRIM_SPHERE_BIG) big_sphere_roots(F3(g), g.w, o, d, t1, t2);
        else {
            f3 co = F3(g) - o;
            const float b = dot(d, co), s = sqrt_fast(fmaf(b, b, -fmaf(co.z, co.z, fmaf(co.y, co.y, fmaf(co.x, co.x, -g.w)))));
            t1 = b - s; t2 = b + s;
        }
        bool near_behind = t1 < tmin;
        float tc = near_behind ? t2 : t1;
        bool ge = tc >= tmin;
        bvh_take(H, ge, tc, q);
        const bool cand = ge & near_behind & (q > H.steal);
        H.steal = cand ? q : H.steal; H.steal_t = cand ? tc : H.steal_t;
    } else if (is_aaplane_type(type)) {
        const float oa = type == PRIM_AAPLANE_X ? o.x : (type == PRIM_AAPLANE_Y ? o.y : o.z);
        const float ia = type == PRIM_AAPLANE_X ? inv_d.x : (type == PRIM_AAPLANE_Y ? inv_d.y : inv_d.z);
        float t = (M.ld1(q) - oa) * ia;
        bvh_take(H, t >= tminp, t, q);
    } else if (type == PRIM_PLANE) {
        float4 g = M.ld4(q);
        float t = (g.w - dot(F3(g), o)) * rcp_fast(dot(F3(g), d));
        bvh_take(H, t >= tminp, t, q);
    } else if (is_strict_type(type)) {      // the reference's own operation sequence (rt_device.h PRIM_*_STRICT): same bits as closest_hit_runs
        float t;
        const bool ok = strict_slot<false>(M, type, q, o, strict_dir(d), tminp, t);
        bvh_take(H, ok, t, q);
    } else {
        float4 g = M.ld4(q), g1 = M.ld4(q + SLOT_G1), g2 = M.ld4(q + SLOT_G2);
        float t = (g.w - dot(F3(g), o)) * rcp_fast(dot(F3(g), d));
        f3 p = madd(d, t, o);
        float x = fmaf(g1.z, p.z, fmaf(g1.y, p.y, fmaf(g1.x, p.x, g1.w)));
        float y = fmaf(g2.z, p.z, fmaf(g2.y, p.y, fmaf(g2.x, p.x, g2.w)));
        bool in;
        if (type == PRIM_RECT) { float4 g3 = M.ld4(q + SLOT_G3); in = (fabsf(x) <= g3.x) & (fabsf(y) <= g3.y); }
        else                   in = (x >= 0.0f) & (y >= 0.0f) & (x + y <= 1.0f);
        bvh_take(H, (t >= tminp) & in, t, q);
    }
}

// Slab test of one node's (padded) box: entry distance in tn.  fminf/fmaxf drop a NaN operand (0 * inf when the origin lies
// on a slab plane of an axis the ray does not move along).  The 1e-5 slack is far inside the padding of rt_bvh.h.
// (lo - o) * (1/d), not the fused lo * (1/d) - o * (1/d): for a ray nearly parallel to a slab both products are huge and
// their difference loses every digit - measured: 5 of 1.5e8 rays culled a box they hit.
__device__ __forceinline__ bool bvh_box(float4 lo, float4 hi, f3 o, f3 inv_d, float closest, float& tn) {
    float ax = (lo.x - o.x) * inv_d.x, bx = (hi.x - o.x) * inv_d.x;
    float ay = (lo.y - o.y) * inv_d.y, by = (hi.y - o.y) * inv_d.y;
    float az = (lo.z - o.z) * inv_d.z, bz = (hi.z - o.z) * inv_d.z;
    tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return (tn <= tf * 1.00001f) & (tf >= 0.0f) & (tn * 0.99999f <= closest);
}

// The scene query.  stack: this lane's entries in LDS, strided by the block size (bank-conflict free); the tree's depth
// bounds how many it needs (launch_path_bvh sizes the allocation from it).  "While-while" walk: every lane first descends
// through inner nodes until it holds a leaf (or nothing), then the wave tests leaves together, so lanes at inner nodes and
// lanes at leaves do not serialise each other's code on every step.
#define RT_BVH_NONE 0xFFFFFFFFU
__device__ __forceinline__ uint32_t closest_hit_bvh(const DevScene& SD, const SceneMem<false> M, uint32_t* __restrict__ stack,
                                                    f3 o, f3 d, float tmin, float tminp, float tmax, float& t_out) {
    BvhHit H; H.closest = tmax; H.best = 0U; H.steal = 0U; H.steal_t = 0.0f;
    const f3 inv_d = F3(rcp_fast(d.x), rcp_fast(d.y), rcp_fast(d.z));
    const uint32_t* __restrict__ unb = reinterpret_cast<const uint32_t*>(SD.blob + SD.off_bvh_unbounded);
    for (int i = 0; i < SD.n_bvh_unbounded; ++i) bvh_test(M, unb[i], o, d, inv_d, tmin, tminp, H);
    if (SD.n_bvh_refs > 0) {
        const float4* __restrict__ nodes = SD.blob + SD.off_bvh_nodes;
        const uint32_t* __restrict__ refs = reinterpret_cast<const uint32_t*>(SD.blob + SD.off_bvh_refs);
        int sp = 0;
        uint32_t cur = 0U;                 // the root, an inner node
        // a walk visits every node at most once; the budget and the stack guard only matter if the tree were malformed
        // (they turn an endless loop or an out-of-range LDS write into a wrong pixel)
        int budget = 2 * SD.n_bvh_nodes + 2;
        for (;;) {
            while (!(cur & 0x80000000U) && --budget > 0) {    // inner: one 64-byte fetch holds both children's boxes
                const float4* __restrict__ nd = nodes + 4U * cur;
                const float4 l0 = nd[0], l1 = nd[1], r0 = nd[2], r1 = nd[3];
                const uint32_t rl = __float_as_uint(l0.w), rr = __float_as_uint(l1.w);
                float tl, tr;
                const bool hl = bvh_box(l0, l1, o, inv_d, H.closest, tl), hr = bvh_box(r0, r1, o, inv_d, H.closest, tr);
                if (hl & hr & (sp < SD.bvh_depth)) {
                    const bool left_first = tl <= tr;
                    stack[sp * 256] = left_first ? rr : rl; ++sp;
                    cur = left_first ? rl : rr;
                } else if (hl | hr) cur = hl ? rl : rr;
                else if (sp == 0) cur = RT_BVH_NONE;
                else { --sp; cur = stack[sp * 256]; }
            }
            if (cur == RT_BVH_NONE || budget <= 0) break;
            const uint32_t first = (cur & 0x7fffffffU) >> 3, n = (cur & 7U) + 1U;
            for (uint32_t i = 0; i < n; ++i) bvh_test(M, refs[first + i], o, d, inv_d, tmin, tminp, H);
            if (sp == 0) break;
            --sp; cur = stack[sp * 256];
        }
    }
    if ((H.steal > H.best) & (H.steal_t > H.closest)) H.best = H.steal;
    t_out = H.closest;
    return H.best;
}

// The bounce loop of path_body with the tree walk as its scene query (one ray per lane, the walk as a function); the scene is
// read from global memory (per-lane addresses).  Kept under debug flag 32 as the A/B reference of the scheduled form below.
template <int VARIANT>
__device__ __forceinline__ void path_bvh_body_simple(const DevScene& SD, const PathArgs& A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    extern __shared__ uint32_t stack_mem[];       // (tree depth) x 256 entries
    uint32_t* stack = stack_mem + threadIdx.x;
    const SceneMem<false> M{reinterpret_cast<const char*>(SD.blob)};
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
            ++L.rays;
            float t;
            uint32_t hit = closest_hit_bvh(SD, M, stack, L.o, L.d, A.t_min, A.t_min_pos, A.t_max, t);
            shade<VARIANT, false>(M, A, 0U, hit == 0U, hit, t, L);
        }
    }
    if (inside) out[(size_t)slice * A.width * A.height + pixel] = make_float4(L.acc.x, L.acc.y, L.acc.z, (A.debug_flags & 4) ? (float)L.rays : (float)L.n_started);
    wave_add_u64(counters + 2 * ((blockIdx.x + blockIdx.y) % RT_COUNTER_BANKS), L.rays);
}


// ---- k_path_bvh, scheduled form -----------------------------------------------------------------------------------------
// With one ray per lane and the walk as a function (path_bvh_body_simple below), a wave spends most of its time waiting for
// its slowest lane: a ray needs 22 inner steps and 2.6 leaves on average on the stress scene, the slowest of 64 several times
// that.  Here a lane is a small state machine - START (ray ready, walk not begun), TRAV (holds an inner node), LEAF (holds a
// leaf), SHADE (walk finished), IDLE (path ended) - and the wave runs a STAGE when enough lanes wait for it: several rounds of
// inner steps per iteration, leaves and shading when a quarter of the wave has queued up for them or nothing else can run,
// regeneration likewise.  A lane never waits for another lane's walk, only for company in its next stage.  Per lane the
// sequence of operations - and so every bit of the result - is that of the simple form.
enum { BVH_IDLE = 0, BVH_START = 1, BVH_TRAV = 2, BVH_LEAF = 3, BVH_SHADE = 4 };
#define RT_BVH_INNER_ROUNDS 8
#define RT_BVH_QUORUM 48

template <int VARIANT>
__device__ __forceinline__ void path_bvh_body(const DevScene& SD, const PathArgs& A, float4* __restrict__ out, unsigned long long* __restrict__ counters) {
    extern __shared__ uint32_t stack_mem[];       // (tree depth) x 256 entries
    uint32_t* stack = stack_mem + threadIdx.x;
    const SceneMem<false> M{reinterpret_cast<const char*>(SD.blob)};
    const float4* __restrict__ nodes = SD.blob + SD.off_bvh_nodes;
    const uint32_t* __restrict__ refs = reinterpret_cast<const uint32_t*>(SD.blob + SD.off_bvh_refs);
    const uint32_t* __restrict__ unb = reinterpret_cast<const uint32_t*>(SD.blob + SD.off_bvh_unbounded);
    int px, py;
    bool inside = pixel_of_thread(A.width, A.height, px, py);
    const int slice = blockIdx.y;
    int s_first = A.sample_begin + slice * A.slice_len;
    const int s_end = min(s_first + A.slice_len, A.sample_end);
    if (!inside) s_first = s_end;
    const uint32_t pixel = (uint32_t)(py * A.width + px);
    const uint32_t pkey = rng_pixel_key(A.seed, pixel);
    PathLane L = lane_init(s_first);
    int state = BVH_IDLE, sp = 0, budget = 0;
    uint32_t cur = 0U;
    BvhHit H; H.closest = 0.0f; H.best = 0U; H.steal = 0U; H.steal_t = 0.0f;
    f3 inv_d = F3(0, 0, 0);

    for (;;) {
        const bool want = state == BVH_IDLE && L.s_next < s_end;
        const unsigned long long m_want = __builtin_amdgcn_ballot_w64(want);
        const int n_trav = __popcll(__builtin_amdgcn_ballot_w64(state == BVH_TRAV || state == BVH_START));
        const int n_leaf = __popcll(__builtin_amdgcn_ballot_w64(state == BVH_LEAF));
        const int n_shade = __popcll(__builtin_amdgcn_ballot_w64(state == BVH_SHADE));
        const int n_want = __popcll(m_want);
        if (n_trav + n_leaf + n_shade + n_want == 0) break;

        // regeneration: idle lanes take the next sample of their pixel
        if (n_want > 0 && (n_want >= RT_BVH_QUORUM || n_trav + n_leaf + n_shade < RT_BVH_QUORUM)) {
            regenerate(A, px, py, pkey, want, m_want, 0ULL, L);
            if (want) state = BVH_START;
        }
        // inner steps (a ray that is ready first tests the planes and takes the root)
        for (int round = 0; round < RT_BVH_INNER_ROUNDS; ++round) {
            if (__builtin_amdgcn_ballot_w64(state == BVH_TRAV || state == BVH_START) == 0ULL) break;
            if (state == BVH_START) {
                ++L.rays;
                H.closest = A.t_max; H.best = 0U; H.steal = 0U; H.steal_t = 0.0f;
                inv_d = F3(rcp_fast(L.d.x), rcp_fast(L.d.y), rcp_fast(L.d.z));
                for (int i = 0; i < SD.n_bvh_unbounded; ++i) bvh_test(M, unb[i], L.o, L.d, inv_d, A.t_min, A.t_min_pos, H);
                sp = 0; cur = 0U; budget = 2 * SD.n_bvh_nodes + 2;
                state = SD.n_bvh_refs > 0 ? BVH_TRAV : BVH_SHADE;
            } else if (state == BVH_TRAV) {
                const float4* __restrict__ nd = nodes + 4U * cur;
                const float4 l0 = nd[0], l1 = nd[1], r0 = nd[2], r1 = nd[3];
                const uint32_t rl = __float_as_uint(l0.w), rr = __float_as_uint(l1.w);
                float tl, tr;
                const bool hl = bvh_box(l0, l1, L.o, inv_d, H.closest, tl), hr = bvh_box(r0, r1, L.o, inv_d, H.closest, tr);
                if (hl & hr & (sp < SD.bvh_depth)) {
                    const bool left_first = tl <= tr;
                    stack[sp * 256] = left_first ? rr : rl; ++sp;
                    cur = left_first ? rl : rr;
                } else if (hl | hr) cur = hl ? rl : rr;
                else if (sp == 0) cur = RT_BVH_NONE;
                else { --sp; cur = stack[sp * 256]; }
                if (cur == RT_BVH_NONE || --budget <= 0) state = BVH_SHADE;      // (the budget only matters for a malformed tree)
                else if (cur & 0x80000000U) state = BVH_LEAF;
            }
        }
        // leaves
        {
            const int nl = __popcll(__builtin_amdgcn_ballot_w64(state == BVH_LEAF));
            const int nt = __popcll(__builtin_amdgcn_ballot_w64(state == BVH_TRAV || state == BVH_START));
            if (nl > 0 && (nl >= RT_BVH_QUORUM || nt < RT_BVH_QUORUM)) {
                if (state == BVH_LEAF) {
                    const uint32_t first = (cur & 0x7fffffffU) >> 3, n = (cur & 7U) + 1U;
                    for (uint32_t i = 0; i < n; ++i) bvh_test(M, refs[first + i], L.o, L.d, inv_d, A.t_min, A.t_min_pos, H);
                    if (sp == 0) state = BVH_SHADE;
                    else { --sp; cur = stack[sp * 256]; state = (cur & 0x80000000U) ? BVH_LEAF : BVH_TRAV; }
                }
            }
        }
        // shading
        {
            const int ns = __popcll(__builtin_amdgcn_ballot_w64(state == BVH_SHADE));
            const int nb = __popcll(__builtin_amdgcn_ballot_w64(state == BVH_TRAV || state == BVH_START || state == BVH_LEAF));
            if (ns > 0 && (ns >= RT_BVH_QUORUM || nb < RT_BVH_QUORUM)) {
                if (state == BVH_SHADE) {
                    if ((H.steal > H.best) & (H.steal_t > H.closest)) H.best = H.steal;
                    shade<VARIANT, false>(M, A, 0U, H.best == 0U, H.best, H.closest, L);
                    state = L.alive ? BVH_START : BVH_IDLE;
                }
            }
        }
    }
    if (inside) out[(size_t)slice * A.width * A.height + pixel] = make_float4(L.acc.x, L.acc.y, L.acc.z, (A.debug_flags & 4) ? (float)L.rays : (float)L.n_started);
    wave_add_u64(counters + 2 * ((blockIdx.x + blockIdx.y) % RT_COUNTER_BANKS), L.rays);
}

} // namespace rt
