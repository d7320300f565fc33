// rt_strict.h - IEEE-exact device arithmetic for the deterministic examples 01/02.
//
// 01_Basic and 02_SoftShadows are compared with the reference per pixel, and their images are full of
// exact geometric ties (room edges that pass through pixel centres): which wall wins such a tie is decided
// by the last bit of t.  So everything that feeds a DECISION in the Whitted kernels (primary ray, the
// ray/primitive tests, hit point, normals, reflect/refract directions, the 02 light-disk sampler) is written
// here with the reference's own operator order, no FMA contraction, and correctly rounded division and
// square root (hipcc's default for '/' and sqrtf) - the same float operations, bit for bit, that the
// reference's x86 build executes.  Only colour arithmetic downstream of the decisions uses fused ops.
// The path-tracing kernels (03/04), which are compared statistically, use the fast forms of rt_device.h.
#pragma once
#include "rt_device.h"

#pragma clang fp contract(off)

namespace rt {
// a struct rather than a namespace: member lookup then hides the fast rt::dot/normalize/... overloads from ADL
struct strict {

static __device__ __forceinline__ f3 add(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
static __device__ __forceinline__ f3 sub(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
static __device__ __forceinline__ f3 muls(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
static __device__ __forceinline__ f3 mulv(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
static __device__ __forceinline__ f3 mix(f3 a, f3 b, float r) { return add(a, muls(sub(b, a), r)); }                 // math::Mix (Math.hpp:323-326): a + ((b - a) * ratio)
static __device__ __forceinline__ float dot(f3 a, f3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }          // Math.hpp:261-264
static __device__ __forceinline__ f3 cross(f3 a, f3 b) {                                                               // Math.hpp:274-281
    return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static __device__ __forceinline__ float length(f3 v) { return sqrtf((v.x * v.x) + (v.y * v.y) + (v.z * v.z)); }       // Math.hpp:195-198
static __device__ __forceinline__ f3 normalize(f3 v) {                                                                 // Math.hpp:233-238
    float len = length(v);
    if (len > 0) return F3(v.x / len, v.y / len, v.z / len);
    return F3(0, 0, 0);
}
static __device__ __forceinline__ f3 reflect(f3 v, f3 n) { return sub(v, muls(muls(n, 2.0f), dot(v, n))); }          // Math.hpp:291-294
static __device__ __forceinline__ f3 refract(f3 v, f3 n, float eta) {                                                  // Math.hpp:306-312
    float d = dot(v, n);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return reflect(v, n);
    return sub(muls(v, eta), muls(n, eta * d + sqrtf(k)));
}

// Ray::intersectsSphere, literally (Common/Ray.hpp:94-137); g = (centre, radius)
static __device__ __forceinline__ bool hit_sphere(float4 g, f3 o, f3 d, float tmin, float tmax, float& t) {
    f3 oc = sub(o, F3(g));
    float a = dot(d, d);
    float b = 2.0f * dot(d, oc);
    float c = dot(oc, oc) - (g.w * g.w);
    float disc = b * b - 4.0f * a * c;
    if (disc < 0) return false;
    float sq = sqrtf(disc);
    float t1 = (-b - sq) / (2.0f * a);
    float t2 = (-b + sq) / (2.0f * a);
    if (t1 < tmin && t2 < tmin) return false;
    if (t1 < tmin) t1 = tmax;
    if (t2 < tmin) t2 = tmax;
    float tr = (t2 < t1) ? t2 : t1;
    t = tr;
    return tr <= tmax;
}
// Ray::intersectsPlane, literally (Ray.hpp:223-249); n = unit normal, p0 = point
static __device__ __forceinline__ bool hit_plane(f3 n, f3 p0, f3 o, f3 d, float tmin, float tmax, float& t) {
    float dn = dot(d, n);
    if (dn != 0) {
        t = ((n.x * p0.x - n.x * o.x) + (n.y * p0.y - n.y * o.y) + (n.z * p0.z - n.z * o.z)) / dn;
        return t > 0 && t >= tmin && t <= tmax;
    }
    return false;
}
// Ray::intersectsTriangle (Ray.hpp:150-212) with normal and Inverse([e2 e1 n]) rows 0,1 precomputed on the host by
// the same float operations; g = {(n, -), (v0, -), (inv row 0, -), (inv row 1, -)}
static __device__ __forceinline__ bool hit_tri(const float4* __restrict__ g, f3 o, f3 d, float tmin, float tmax, float& t) {
    f3 n = F3(g[0]), v0 = F3(g[1]), r0 = F3(g[2]), r1 = F3(g[3]);
    float dn = dot(d, n);
    if (dn != 0) {
        t = ((n.x * v0.x - n.x * o.x) + (n.y * v0.y - n.y * o.y) + (n.z * v0.z - n.z * o.z)) / dn;
        if (t > 0 && t >= tmin && t <= tmax) {
            f3 pt = sub(add(o, muls(d, t)), v0);
            float x = r0.x * pt.x + r0.y * pt.y + r0.z * pt.z;
            float y = r1.x * pt.x + r1.y * pt.y + r1.z * pt.z;
            return x >= 0.0f && y >= 0.0f && x + y <= 1.0f;
        }
    }
    return false;
}

// Whitted scene layout: planes carry a second float4 with the point p0 (geom stride 2), see rt_api.hip.
// Closest hit of 01/02 TraceTay (01 Main.cpp:389-403): same [tMin,tMax] for every element, strict t < nearest.
static __device__ __forceinline__ int closest_hit(const Scn& S, f3 o, f3 d, float tmin, float tmax, float& t_out) {
    int best = -1;
    float nearest = 3.402823466e+38f;
    for (int r = 0; r < S.n_runs; ++r) {
        int4 run = S.runs[r];
        const float4* g = S.geom + run.y;
        if (run.x == PRIM_SPHERE) {
            for (int i = 0; i < run.z; ++i) { float t; if (hit_sphere(g[i], o, d, tmin, tmax, t) && t < nearest) { nearest = t; best = run.w + i; } }
        } else if (run.x == PRIM_PLANE) {
            for (int i = 0; i < run.z; ++i) { float t; if (hit_plane(F3(g[2 * i]), F3(g[2 * i + 1]), o, d, tmin, tmax, t) && t < nearest) { nearest = t; best = run.w + i; } }
        } else if (run.x == PRIM_TRI) {
            for (int i = 0; i < run.z; ++i) { float t; if (hit_tri(g + 4 * i, o, d, tmin, tmax, t) && t < nearest) { nearest = t; best = run.w + i; } }
        }
    }
    t_out = nearest;
    return best;
}
// Shadow form (outColor == nullptr): any element inside [tMin,tMax].
static __device__ __forceinline__ bool any_hit(const Scn& S, f3 o, f3 d, float tmin, float tmax) {
    bool hit = false;
    for (int r = 0; r < S.n_runs; ++r) {
        int4 run = S.runs[r];
        const float4* g = S.geom + run.y;
        if (run.x == PRIM_SPHERE) {
            for (int i = 0; i < run.z; ++i) { float t; hit |= hit_sphere(g[i], o, d, tmin, tmax, t); }
        } else if (run.x == PRIM_PLANE) {
            for (int i = 0; i < run.z; ++i) { float t; hit |= hit_plane(F3(g[2 * i]), F3(g[2 * i + 1]), o, d, tmin, tmax, t); }
        } else if (run.x == PRIM_TRI) {
            for (int i = 0; i < run.z; ++i) { float t; hit |= hit_tri(g + 4 * i, o, d, tmin, tmax, t); }
        }
    }
    return hit;
}

// Render's primary ray of 01/02 (01 Main.cpp:340-344): ((2(i+.5))/w - 1) * tan * w / h, left to right
static __device__ __forceinline__ f3 primary_dir(int px, int py, float w, float h, float tan_half) {
    float x = (2.0f * ((float)px + 0.5f) / w - 1.0f) * tan_half * w / h;
    float y = -(2.0f * ((float)py + 0.5f) / h - 1.0f) * tan_half;
    return normalize(F3(x, y, -1.0f));
}

// RandomVectorToLightSphere (02 Main.cpp:539-566): the normalised direction
// to a random point of the light's disk; u = {vBias x,y,z draws, rBias draw} in [0,1)
static __device__ __forceinline__ f3 to_light_disk(f3 p, f3 lp, float radius, float u0, float u1, float u2, float u3) {
    f3 to = normalize(sub(lp, p));
    f3 vb = F3(2.0f * u0 + -1.0f, 2.0f * u1 + -1.0f, 2.0f * u2 + -1.0f);
    f3 pl = normalize(cross(to, add(to, vb)));
    f3 target = add(lp, muls(muls(pl, radius), u3));
    return normalize(sub(target, p));
}

};
} // namespace rt

#pragma clang fp contract(fast)

namespace rt {
// 02's shadow samples (MAX_SHADOW_SAMPLES per shaded hit, 02 Main.cpp:452-467) are the one place of the Whitted kernels
// where the fast forms are used: a sample only moves the pixel by 1/samples of one light's contribution and its
// direction is random anyway, so a last-bit difference in it cannot create the systematic edge errors that made the
// closest-hit arithmetic exact; the ray COUNT does not depend on the outcome at all.  Same layout as strict::any_hit.
struct fastshadow {
static __device__ __forceinline__ bool any_hit(const Scn& S, f3 o, f3 d, float tmin, float tmax) {
    bool hit = false;
    for (int r = 0; r < S.n_runs; ++r) {
        int4 run = S.runs[r];
        const float4* g = S.geom + run.y;
        if (run.x == PRIM_SPHERE) {
            for (int i = 0; i < run.z; ++i) {   // Ray::intersectsSphere with a = d.d = 1: roots b -+ s
                float4 q = g[i];
                f3 co = F3(q) - o;
                float b = dot(d, co);
                float c = fmaf(co.z, co.z, fmaf(co.y, co.y, fmaf(co.x, co.x, -q.w * q.w)));
                float s = sqrt_fast(fmaf(b, b, -c));
                float t1 = b - s, t2 = b + s;
                bool nb = t1 < tmin;
                float tc = nb ? t2 : t1;
                hit |= (tc >= tmin) & (nb | (tc <= tmax));
            }
        } else if (run.x == PRIM_PLANE) {
            for (int i = 0; i < run.z; ++i) {
                f3 n = F3(g[2 * i]), p0 = F3(g[2 * i + 1]);
                float t = dot(n, p0 - o) * rcp_fast(dot(d, n));
                hit |= (t > 0.0f) & (t >= tmin) & (t <= tmax);
            }
        } else if (run.x == PRIM_TRI) {
            for (int i = 0; i < run.z; ++i) { float t; hit |= strict::hit_tri(g + 4 * i, o, d, tmin, tmax, t); }
        }
    }
    return hit;
}
// The same with the four uniforms given as their 16-bit integers k (u = k/65536) and radius_s = radius/65536: identical
// results (scaling by a power of two is exact, so each product rounds the same real number), four multiplies fewer; and the
// two Normalize zero guards become "+ 1e-30" on the squared length: a zero vector still maps to zero, and a non-zero squared
// length here (>= ~1e-14) is the same float after adding 1e-30.
static __device__ __forceinline__ f3 to_light_disk_raw(f3 p, f3 lp, f3 to, float radius_s, float k0, float k1, float k2, float k3) {
    f3 vb = F3(fmaf(k0, 2.0f / 65536.0f, -1.0f), fmaf(k1, 2.0f / 65536.0f, -1.0f), fmaf(k2, 2.0f / 65536.0f, -1.0f));
    f3 pl = cross(to, to + vb);
    pl = pl * rsq_fast(dot(pl, pl) + 1e-30f);
    f3 dir = madd(pl, radius_s * k3, lp) - p;
    return dir * rsq_fast(dot(dir, dir) + 1e-30f);
}
// RandomVectorToLightSphere (02 Main.cpp:539-566); 'to' = normalize(lp - p) is the same for every sample of a light
static __device__ __forceinline__ f3 to_light_disk(f3 p, f3 lp, f3 to, float radius, float u0, float u1, float u2, float u3) {
    f3 vb = F3(fmaf(2.0f, u0, -1.0f), fmaf(2.0f, u1, -1.0f), fmaf(2.0f, u2, -1.0f));
    f3 pl = normalize(cross(to, to + vb));
    f3 target = madd(pl, radius * u3, lp);
    return normalize(target - p);
}
};
} // namespace rt
