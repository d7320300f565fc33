// rt_device.h - device-side scene layout and the small float3 toolkit shared by the kernels.
// gfx950 only.  Reference semantics cited per function (paths relative to /root/reference/Sources).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

// ---- flattened, precomputed scene (built on the host by compile_scene, rt_api.hip) -------------
// One contiguous blob of float4 records; sections at float4 offsets off_*:
// runs : one int4 per maximal run of same-type primitives, IN SCENE ORDER (order matters: ties
//        and the sphere tMax rule depend on it) = {type, first geom float4 index, count, first prim index}
// geom : per-type records, float4 units (index relative to off_geom):
//        sphere   1: (cx, cy, cz, r*r)               [Whitted scenes: (cx, cy, cz, r)]
//        plane    1: (nx, ny, nz, n.p0)              [Whitted scenes: 2: (n, 0) (p0, 0), see rt_strict.h]
//        aaplane  1: (p0[axis], n[axis] = +-1, 0, 0)  path scenes only: a plane whose unit normal is +-e_axis
//        rect     5: (R00,R01,R02,px) (R10,R11,R12,py) (R20,R21,R22,pz) (hw,hh,nwx,nwy) (nwz,0,0,0)
//                    R = GetRotationMat(-orientation), nw = GetRotationMat(orientation)*(0,0,1)
//        triangle 4: (nx,ny,nz,n.v0) (v0x,v0y,v0z,0) (i00,i01,i02,0) (i10,i11,i12,0)
//                    n = normalize(cross(e2,e1)), i = rows 0,1 of Inverse([e2 e1 n])
// info : one int4 per primitive = {device type, geom index, material index, flags}
// mats : 3 float4 per material:
//        m0 (albedo.rgb, type as int bits)   m1 (specularIntensity, shininess, p2s, r2r)
//        m2 (refractionEta, cone, refractionIndex, schlick r0)
//        cone: variant 04: cos(thetaMax/57.2958); variant 03: thetaMax/57.2958 (radians), <0 = no sampling
// lights: 2 float4 per light: (pos.xyz, radius) (color.rgb, 0)
struct DevScene {
    const float4* __restrict__ blob;
    int off_runs, off_geom, off_info, off_mats, off_lights, total_f4;
    int n_runs, n_prims, n_mats, n_lights;
    int has_aa;     // any axis-aligned plane run (then 1/d is computed once per ray)
};

// Section pointers over either the global blob (wave-uniform reads become scalar loads) or its LDS copy
// (reads become ds_read: operands arrive in VGPRs, which is what the VALU wants on gfx950 - an SGPR
// operand halves the issue rate of a VALU instruction, profiles/r01_valu_issue_rates_ubench.txt).
struct Scn {
    const int4* runs; const float4* geom; const int4* info; const float4* mats; const float4* lights;
    int n_runs, n_lights;
};
__device__ __forceinline__ Scn scn_over(const DevScene& S, const float4* base) {
    Scn v;
    v.runs = reinterpret_cast<const int4*>(base + S.off_runs); v.geom = base + S.off_geom;
    v.info = reinterpret_cast<const int4*>(base + S.off_info); v.mats = base + S.off_mats; v.lights = base + S.off_lights;
    v.n_runs = S.n_runs; v.n_lights = S.n_lights;
    return v;
}

enum { PRIM_SPHERE = 0, PRIM_PLANE = 1, PRIM_RECT = 2, PRIM_TRI = 3, PRIM_AAPLANE_X = 4, PRIM_AAPLANE_Y = 5, PRIM_AAPLANE_Z = 6 };
enum { MAT_PHONG = 0, MAT_DIFFUSE = 1, MAT_METAL = 2, MAT_DIELECTRIC = 3, MAT_LIGHT = 4, MAT_NONE = 5 };

// ---- float3 -----------------------------------------------------------------------------------
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 F3(float4 v) { return F3(v.x, v.y, v.z); }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return F3(fmaf(a.y, b.z, -a.z * b.y), fmaf(a.z, b.x, -a.x * b.z), fmaf(a.x, b.y, -a.y * b.x));
}
__device__ __forceinline__ f3 madd(f3 a, float s, f3 b) { return F3(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)); } // a*s+b
__device__ __forceinline__ f3 sel(bool c, f3 a, f3 b) { return F3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }

__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq_fast(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }

// math::Normalize (Common/Math.hpp:233-238): zero vector stays zero.
__device__ __forceinline__ f3 normalize(f3 v) {
    float l2 = dot(v, v);
    float inv = l2 > 0.0f ? rsq_fast(l2) : 0.0f;
    return v * inv;
}
// math::Reflect (Math.hpp:291-294)
__device__ __forceinline__ f3 reflect(f3 v, f3 n) { return madd(n, -2.0f * dot(v, n), v); }
// math::Refract, reflect-on-TIR form (Math.hpp:306-312)
__device__ __forceinline__ f3 refract(f3 v, f3 n, float eta) {
    float dn = dot(v, n);
    float k = fmaf(-eta * eta, fmaf(-dn, dn, 1.0f), 1.0f);
    f3 refl = madd(n, -2.0f * dn, v);
    float s = fmaf(eta, dn, sqrt_fast(fmaxf(k, 0.0f)));
    f3 refr = madd(n, -s, v * eta);
    return sel(k < 0.0f, refl, refr);
}

// ---- counter-based RNG (DESIGN.md "RNG"): every draw = f(seed, pixel, sample, counter) ----------
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
__device__ __forceinline__ uint32_t rng_pixel_key(uint32_t seed, uint32_t pixel) { return lowbias32(pixel + 0x9E3779B9U * seed + 0x85EBCA6BU); }
__device__ __forceinline__ uint32_t rng_sample_key(uint32_t pkey, uint32_t sample) { return lowbias32(pkey ^ (sample * 0x9E3779B1U)); }
__device__ __forceinline__ float rng_u01(uint32_t skey, uint32_t ctr) {
    uint32_t r = lowbias32(skey + (ctr + 1U) * 0x9E3779B9U);
    return (float)(r >> 8) * (1.0f / 16777216.0f);
}

// two 16-bit uniforms from ONE draw: a = low half, b = high half, each k/65536
__device__ __forceinline__ void rng_pair(uint32_t skey, uint32_t ctr, float& a, float& b) {
    uint32_t r = lowbias32(skey + (ctr + 1U) * 0x9E3779B9U);
    a = (float)(r & 0xFFFFU) * (1.0f / 65536.0f);
    b = (float)(r >> 16) * (1.0f / 65536.0f);
}

// ---- ray / primitive tests (fast forms, path tracing kernels) ----------------------------------------
// Ray::intersectsSphere (Common/Ray.hpp:94-137) with a = d.d = 1 (every traced direction is unit); g = (c, r*r).
// Roots t1 <= t2; "a root below tMin becomes tMax" (:127-128): a ray that starts inside a sphere whose far root
// lies beyond tMax still reports a hit AT tMax (:129-134).  disc < 0 gives NaN roots, which fail every compare.
__device__ __forceinline__ bool hit_sphere(float4 g, f3 o, f3 d, float tmin, float tmax, float& t) {
    f3 co = F3(g) - o;
    float b = dot(d, co);
    float c = fmaf(co.z, co.z, fmaf(co.y, co.y, fmaf(co.x, co.x, -g.w)));
    float s = sqrt_fast(fmaf(b, b, -c));
    float t1 = b - s, t2 = b + s;
    bool near_behind = t1 < tmin;
    float tc = near_behind ? t2 : t1;
    t = fminf(tc, tmax);
    return (tc >= tmin) & (near_behind | (tc <= tmax));
}
// Ray::intersectsPlane (Ray.hpp:223-249); g = (n, n.p0).  dot==0 gives inf/NaN, which fails the range test.
// tminp = max(tMin, smallest positive float) folds the reference's "t > 0 && t >= tMin" into one compare.
__device__ __forceinline__ bool hit_plane(float4 g, f3 o, f3 d, float tminp, float tmax, float& t) {
    f3 n = F3(g);
    t = (g.w - dot(n, o)) * rcp_fast(dot(d, n));
    return (t >= tminp) & (t <= tmax);
}
// The same for a plane whose normal is +-e_axis: n.p0 - n.o = +-(p0_a - o_a), d.n = +-d_a, so
// t = (p0_a - o_a) * (1/d_a) with 1/d shared by all such planes of a ray; bit-identical to hit_plane.
__device__ __forceinline__ bool hit_aaplane(float p0a, float oa, float inv_da, float tminp, float tmax, float& t) {
    t = (p0a - oa) * inv_da;
    return (t >= tminp) & (t <= tmax);
}
// scene::Rectangle::intersectsRay (04 Scene/Rectangle.hpp:51-97) over Ray::intersectsAARectangleXy
// (Ray.hpp:263-283): ray into object space by R(-orient), z=0 plane, inclusive |x|<=hw, |y|<=hh.
__device__ __forceinline__ bool hit_rect(const float4* g, f3 o, f3 d, float tminp, float tmax, float& t) {
    float4 r0 = g[0], r1 = g[1], r2 = g[2], r3 = g[3];
    f3 oc = o - F3(r0.w, r1.w, r2.w);
    float oz = dot(F3(r2), oc), dz = dot(F3(r2), d);
    t = -oz * rcp_fast(dz);
    float x = fmaf(dot(F3(r0), d), t, dot(F3(r0), oc));
    float y = fmaf(dot(F3(r1), d), t, dot(F3(r1), oc));
    return (t >= tminp) & (t <= tmax) & (fabsf(x) <= r3.x) & (fabsf(y) <= r3.y);
}
// Ray::intersectsTriangle (Ray.hpp:150-212): plane hit, then coordinates in the [e2 e1 n] basis,
// inclusive x>=0, y>=0, x+y<=1.
__device__ __forceinline__ bool hit_tri(const float4* g, f3 o, f3 d, float tminp, float tmax, float& t) {
    float4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
    f3 n = F3(g0);
    t = (g0.w - dot(n, o)) * rcp_fast(dot(d, n));
    f3 pt = madd(d, t, o) - F3(g1);
    float x = dot(F3(g2), pt), y = dot(F3(g3), pt);
    return (t >= tminp) & (t <= tmax) & (x >= 0.0f) & (y >= 0.0f) & (x + y <= 1.0f);
}

// Closest hit, 03 Scene::intersectsRay / 04 scene::List::intersectsRay (03 Utils.h:255-271, 04 Utils.h:362-390):
// tMax shrinks to the closest hit so far and a test accepts t <= tMax, so the LAST element wins exact ties.
__device__ __forceinline__ int closest_hit_shrinking(const Scn& S, bool has_aa, f3 o, f3 d, float tmin, float tminp, float tmax, float& t_out) {
    int best = -1;
    float closest = tmax;
    f3 inv_d = F3(0, 0, 0);
    if (has_aa) inv_d = F3(rcp_fast(d.x), rcp_fast(d.y), rcp_fast(d.z));
    for (int r = 0; r < S.n_runs; ++r) {
        int4 run = S.runs[r];
        const float4* g = S.geom + run.y;
        if (run.x == PRIM_SPHERE) {
            for (int i = 0; i < run.z; ++i) { float t; if (hit_sphere(g[i], o, d, tmin, closest, t)) { closest = t; best = run.w + i; } }
        } else if (run.x == PRIM_AAPLANE_X) {
            for (int i = 0; i < run.z; ++i) { float t; if (hit_aaplane(g[i].x, o.x, inv_d.x, tminp, closest, t)) { closest = t; best = run.w + i; } }
        } else if (run.x == PRIM_AAPLANE_Y) {
            for (int i = 0; i < run.z; ++i) { float t; if (hit_aaplane(g[i].x, o.y, inv_d.y, tminp, closest, t)) { closest = t; best = run.w + i; } }
        } else if (run.x == PRIM_AAPLANE_Z) {
            for (int i = 0; i < run.z; ++i) { float t; if (hit_aaplane(g[i].x, o.z, inv_d.z, tminp, closest, t)) { closest = t; best = run.w + i; } }
        } else if (run.x == PRIM_PLANE) {
            for (int i = 0; i < run.z; ++i) { float t; if (hit_plane(g[i], o, d, tminp, closest, t)) { closest = t; best = run.w + i; } }
        } else if (run.x == PRIM_RECT) {
            for (int i = 0; i < run.z; ++i) { float t; if (hit_rect(g + 5 * i, o, d, tminp, closest, t)) { closest = t; best = run.w + i; } }
        } else {
            for (int i = 0; i < run.z; ++i) { float t; if (hit_tri(g + 4 * i, o, d, tminp, closest, t)) { closest = t; best = run.w + i; } }
        }
    }
    t_out = closest;
    return best;
}

// wave64 sum of a per-lane counter, then one atomic per wave
__device__ __forceinline__ void wave_add_u64(unsigned long long* dst, unsigned int v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, (unsigned long long)v);
}

// 16x16-pixel tile per 256-thread block, 8x8 per wave
__device__ __forceinline__ bool pixel_of_thread(int width, int height, int& px, int& py) {
    int tiles_x = (width + 15) >> 4;
    int tile = blockIdx.x, tx = tile % tiles_x, ty = tile / tiles_x;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    px = tx * 16 + (wave & 1) * 8 + (lane & 7);
    py = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    return px < width && py < height;
}

} // namespace rt
