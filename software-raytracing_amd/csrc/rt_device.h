// rt_device.h - device-side scene layout and the small float3 toolkit shared by the kernels.
// gfx950 only.  Reference semantics cited per function (paths relative to /root/reference/Sources).
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace rt {

// ---- flattened, precomputed scene (built on the host by compile_scene, rt_api.hip) -------------
// One contiguous blob of float4 records; sections at float4 offsets off_*.  Path scenes (03/04) and 02's shadow samples read
// the SLOT section (see "slot layout" below); the sections described here serve the IEEE-exact Whitted tests of rt_strict.h:
// runs : one int4 per maximal run of same-type primitives, IN SCENE ORDER (order matters: the first element wins ties
//        in 01/02) = {type, first geom float4 index, count, first prim index}
// geom : per-type records, float4 units (index relative to off_geom):
//        sphere   1: (cx, cy, cz, r)
//        plane    2: (n, 0) (p0, 0)
//        triangle 4: (nx,ny,nz,n.v0) (v0x,v0y,v0z,0) (i00,i01,i02,0) (i10,i11,i12,0)
//                    n = normalize(cross(e2,e1)), i = rows 0,1 of Inverse([e2 e1 n])
// info : one int4 per primitive = {device type, geom index, material index, flags}
// mats : 3 float4 per material:
//        m0 (albedo.rgb, type as int bits)   m1 (specularIntensity, shininess, p2s, r2r)
//        m2 (refractionEta [path variant 04: (1 - cone)/65536], cone, refractionIndex, schlick r0)
//        m0.x of a dielectric: 1/refractionIndex
//        cone: variant 04: cos(thetaMax/57.2958); variant 03: thetaMax/57.2958 (radians), <0 = no sampling
// lights: 2 float4 per light: (pos.xyz, radius) (color.rgb, 0)
struct DevScene {
    const float4* __restrict__ blob;
    int off_runs, off_geom, off_info, off_mats, off_lights, total_f4;
    int n_runs, n_prims, n_mats, n_lights;
    int has_aa;     // any axis-aligned plane run
    int off_slots;  // path scenes: float4 offset of the slot section (see "slot layout" below); geom/info unused there
    unsigned long long srun_pack[2];      // the first 8 slot runs as (type | count << 3), 16 bits each; srun_packed = all of them fit
    int srun_packed;
    int whitted_branches;                 // Whitted scenes: some material spawns reflection AND refraction at one hit (needs the LDS stack)
    int off_sruns, n_sruns, slot_bytes;   // the runs over the slot section (the 'runs' above belong to the Whitted layout)
    int off_chunks, n_chunks;   // path scenes: int4 {blob byte offset, bytes, first run, n runs} per <= RT_TILE_BYTES piece of the slots
    // optional BVH (rt_bvh.h, rt_bvh_device.h): float4 offsets of the node pairs, of the leaf entries and of the list of
    // unbounded slots (planes); entries are uint32 {blob byte offset of the slot | device type}
    int off_bvh_nodes, off_bvh_refs, off_bvh_unbounded, n_bvh_refs, n_bvh_unbounded, bvh_depth, n_bvh_nodes;
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

enum { PRIM_SPHERE = 0, PRIM_PLANE = 1, PRIM_RECT = 2, PRIM_TRI = 3, PRIM_AAPLANE_X = 4, PRIM_AAPLANE_Y = 5, PRIM_AAPLANE_Z = 6, PRIM_SPHERE_BIG = 7,
       PRIM_PLANE_STRICT = 8, PRIM_RECT_STRICT = 9, PRIM_TRI_STRICT = 10 };
// PRIM_*_STRICT: a planar slot whose supporting plane coincides with that of ANOTHER slot the reference evaluates with different
// arithmetic (a box face resting on a plane, a rectangle lying in a triangle's plane, ...; found when the scene is compiled,
// rt_api.hip find_coplanar).  In exact arithmetic the two distances are equal; in the reference the rounding of two different
// formulas decides which element a ray sees (04 Scene/Rectangle.hpp:58-70 works in the object frame, Scene/Plane.hpp in the world
// frame), about half of the time against scene order.  The fast slot forms fold the object frame into the plane equation and
// produce an EXACT tie there, which scene order always decides - a different image (a glass box on a floor plane: mean -1.1 %,
// profiles/r02_bias_scan.txt).  Such slots are therefore tested in the reference's own operation sequence (strict_* below: its
// operator order, no contraction, true division and square root), so the same rounding decides.  No reference scene has one.
// PRIM_SPHERE_BIG: a sphere whose radius exceeds RT_BIG_SPHERE_RATIO x tMin (03's ground sphere: r = 100, tMin = 0.001).  Same
// slot, same test, but the square root of the discriminant is refined (sqrt_refined below).  Why: the hardware square root is
// good to 1 ulp but not unbiased (v_sqrt_f32: 14.9 % of results one ulp LOW, 0.01 % high, mean -0.149 ulp; the refined form:
// correctly rounded on all 2^24 samples - tools/sqrt_bias.hip, profiles/r02_sqrt_bias.txt); through t = b -+ s that puts hit points preferentially a few 1e-7 of the distance INSIDE the
// sphere, where the next ray's far root lands just above tMin ("acne": a false self-hit).  A ray leaving the surface at cos(theta)
// from a depth h self-hits when cos(theta) < h/tMin - tMin/(2r): never for small spheres, often for large ones.  The reference's
// correctly rounded sqrtf has no such bias; measured against it (scene queries per sample at tMin = 0.001, one sphere seen at a
// grazing angle, tools/big_sphere_check.py): r = 100: +1.6e-3 with the hardware root, +1.5e-4 refined; r = 1000: +1.4e-2 vs -3.7e-4;
// r = 10: +8e-7 vs -1.1e-4 - so the refined root is used from r/tMin = 3e4 up (03: the ground sphere only; 04 and the stress
// scene: none).
#define RT_BIG_SPHERE_RATIO 30000.0f
// A GIANT sphere (radius > RT_GIANT_SPHERE_RATIO x tMin; r = 300 at 03's tMin): every bounce off or inside it is decided by rounding (c = oc.oc - r*r carries noise of
// r*r*2^-24, i.e. distances of r*2^-25/cos(theta) against tMin), and what the roots are then depends on the last bits of the direction: in scenes that have one, scattered
// directions are normalised in the reference's operation order (PathArgs::norm_dirs; strict_dir), and the sphere normal and the hit point follow it too.  Measured on fuzz family B (tools/fuzz_path.py, profiles/r03_path_fuzz.txt 9):
// a glass ground of r = 2500: +37 % scene queries without, -0.6 % with; r = 10000 ... 50000: +2 ... 3 % -> 0.2 %.  Nothing the reference ships is near it.
#define RT_GIANT_SPHERE_RATIO 300000.0f
#ifdef RT_JIT_NORM_DIRS
#define RT_NORM_DIRS(A) (RT_JIT_NORM_DIRS != 0)
#else
#define RT_NORM_DIRS(A) ((A).norm_dirs != 0)
#endif
// Scenes with PRIM_*_STRICT slots (coplanar elements of different arithmetic): the hit point is o + d*t in TWO roundings, as the reference computes it (PathArgs::strict_point).
// Which of two coinciding faces a ray sees is decided by the rounding of two formulas applied to the ray's origin - a hit point; a fused o + d*t lies closer to the face
// it is on than the reference's, and the decisions downstream stop being the coin flips they are there (a glass box standing on a box: +1.3e-2 scene queries, -1e-4 with it;
// tools/fuzz_path.py family C, profiles/r03_path_fuzz.txt 11).
#ifdef RT_JIT_STRICT_POINT
#define RT_STRICT_POINT(A) (RT_JIT_STRICT_POINT != 0)
#else
#define RT_STRICT_POINT(A) ((A).strict_point != 0)
#endif
// REF (template parameter of the path kernels' device code, default RT_REF_RAYS): the reference's arithmetic for rays and spheres (see strict_sphere_roots);
// a scene-specialised build for RT_ARITH_REFERENCE defines RT_REF_RAYS true before this header (rt_jit.hip).
#ifndef RT_REF_RAYS
#define RT_REF_RAYS false
#endif
__device__ __host__ inline bool is_sphere_type(int t) { return t == PRIM_SPHERE || t == PRIM_SPHERE_BIG; }
__device__ __host__ inline bool is_aaplane_type(int t) { return t >= PRIM_AAPLANE_X && t <= PRIM_AAPLANE_Z; }
__device__ __host__ inline bool is_strict_type(int t) { return t >= PRIM_PLANE_STRICT; }
__device__ __host__ inline bool is_wide_type(int t) { return t == PRIM_RECT || t == PRIM_TRI || t == PRIM_RECT_STRICT || t == PRIM_TRI_STRICT; }   // 8 float4 per slot
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
// sqrt to ~0.5 ulp without a second transcendental: one Newton step on s = x*rsq(x).  x < 0 -> NaN (as sqrt), x = 0 -> 0
// (the 2e-38 keeps rsq finite there and vanishes in x + 2e-38 for every x that matters).
__device__ __forceinline__ float sqrt_refined(float x) {
    const float r = rsq_fast(x + 2e-38f), s = x * r;
    return fmaf(fmaf(-s, s, x), 0.5f * r, s);
}

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

__device__ __forceinline__ uint32_t rng_bits(uint32_t skey, uint32_t ctr) { return lowbias32(skey + (ctr + 1U) * 0x9E3779B9U); }
// two 16-bit uniforms from ONE draw: a = low half, b = high half, each k/65536
__device__ __forceinline__ void rng_pair(uint32_t skey, uint32_t ctr, float& a, float& b) {
    uint32_t r = lowbias32(skey + (ctr + 1U) * 0x9E3779B9U);
    a = (float)(r & 0xFFFFU) * (1.0f / 65536.0f);
    b = (float)(r >> 16) * (1.0f / 65536.0f);
}

// the two 16-bit halves of one draw as floats k (the uniforms are k/65536; callers fold the power-of-two scale into a constant)
__device__ __forceinline__ void rng_pair_raw(uint32_t skey, uint32_t ctr, float& a, float& b) {
    uint32_t r = lowbias32(skey + (ctr + 1U) * 0x9E3779B9U);
    a = (float)(r & 0xFFFFU);
    b = (float)(r >> 16);
}

// ---- path-tracing scenes: slot layout ---------------------------------------------------------------
// The path kernels (03/04) read a second section of the blob, 'slots': the primitives IN SCENE ORDER (order
// matters: exact ties and the sphere tMax rule depend on it), each a fixed number of float4 per device type,
//   [G | A | B | C]            sphere, plane, axis-aligned plane      (4 float4 = 64 bytes)
//   [G | A | B | C | G1 G2 G3 S] rectangle, triangle                   (8 float4 = 128 bytes); S = bounding sphere (centre, R^2)
// G..G3 feed the ray test, A,B,C feed shading and sit at the same offsets for every type, so the closest-hit
// loop only has to remember WHERE the winner is (its byte offset in the blob), not what it is:
//   A = (c.xyz, k)   B = (m.xyz, byte offset of the material record)   C = (f.xyz, material class MAT_* as int bits)
//   normal n = (p - c)*k/|p - c| + m, facing vector (p - c)*k/|p - c| + f; the hit is a back face when dot(d, facing) > 0.
//     sphere: c = centre, k = +-1 ('inverted' negates), m = f = 0: Normalize(p - c)   (03 Sphere.hpp:57-73, 04 Scene/Sphere.hpp:58-74)
//     planes, triangle: c = 0, k = 0, m = f = unit normal               (04 Scene/Plane.hpp:55-66)
//     rectangle: m = GetRotationMat(orientation)*(0,0,1), f = row 2 of GetRotationMat(-orientation): the reference
//                decides the side in object space and rotates the normal out with the OTHER matrix (Rectangle.hpp:58-92)
//   G: sphere (c, r*r) | plane (n, n.p0) | aaplane (p0[axis],-,-,-)
//      rectangle G=(z, z.c) G1=(x, -x.c) G2=(y, -y.c) G3=(hw, hh, -, -) with x,y,z the rows of GetRotationMat(-orientation)
//      triangle  G=(n, n.v0) G1=(i0, -i0.v0) G2=(i1, -i1.v0), n = normalize(cross(e2,e1)), i = rows 0,1 of Inverse([e2 e1 n])
// runs (int4 {device type, -, count, -} per maximal run of one type) are read with scalar loads from the GLOBAL blob
// even when the slots are staged in LDS: they steer uniform control flow, which wants SGPRs, while primitive
// operands want VGPRs (an SGPR operand halves VALU issue rate on gfx950, profiles/r01_valu_issue_rates_ubench.txt).
#define RT_TILE_BYTES 16384   /* k_path_tiled stages the slot section through LDS in pieces of at most this size */
enum { SLOT_BYTES = 64, SLOT_A = 16, SLOT_B = 32, SLOT_C = 48, SLOT_G1 = 64, SLOT_G2 = 80, SLOT_G3 = 96, SLOT_BOUND = 112 };

// Scene memory: either the global blob (base pointer + byte offset; uniform offsets become scalar loads) or its LDS
// copy, where an "offset" IS the 32-bit LDS address (the staging base is folded in once, not added per read).
#define RT_LDS_AS __attribute__((address_space(3)))
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool LDS> struct SceneMem {
    const char* base;
    __device__ __forceinline__ float4 ld4(uint32_t off) const {
        if (LDS) { v4f v = *(const RT_LDS_AS v4f*)(uintptr_t)off; return make_float4(v.x, v.y, v.z, v.w); }
        return *reinterpret_cast<const float4*>(base + off);
    }
    __device__ __forceinline__ float ld1(uint32_t off) const {
        if (LDS) return *(const RT_LDS_AS float*)(uintptr_t)off;
        return *reinterpret_cast<const float*>(base + off);
    }
    // blob byte offset -> what ld4/ld1 take
    __device__ __forceinline__ uint32_t at(uint32_t blob_off) const {
        if (LDS) return (uint32_t)(uintptr_t)(const RT_LDS_AS char*)base + blob_off;
        return blob_off;
    }
};

// ---- planar slots in the reference's own operation sequence (PRIM_*_STRICT) ---------------------------------------------------
// Each returns "the element's test passes for tMax = +inf" and its t; the caller adds t <= closest.  tminp = max(tMin, smallest
// positive float) folds the reference's "t > 0 && t >= tMin".
#pragma clang fp contract(off)
// The reference's rays are normalised by Ray's constructor (Common/Ray.hpp:33-34, math::Normalize Math.hpp:233-238), so the length
// of a direction is 1 to within an ulp THERE; the kernels' directions are unit by construction (reflect / refract / sampling frame),
// to a few 1e-7 and not with the rounding pattern of a division by the length.  Where the last bit of t decides, the strict tests
// therefore see the direction as the reference would hold it: d / |d| in its operation order.
__device__ __forceinline__ f3 strict_dir(f3 d) {
    const float len = sqrtf((d.x * d.x) + (d.y * d.y) + (d.z * d.z));
    return len > 0.0f ? F3(d.x / len, d.y / len, d.z / len) : F3(0.0f, 0.0f, 0.0f);
}
// Ray::intersectsSphere in its own operation order (Common/Ray.hpp:105-121), halved: a = d.d, bh = -b/2 = -(d.oc), c = oc.oc - r*r, dq = disc/4 = bh*bh - a*c,
// roots (bh -+ sqrt(dq))/a: the reference's numbers exactly (scaling by 2 and 4 is exact in binary floating point).  Why the order matters: c cancels from terms of size
// |oc|^2 to (nearly) nothing for a ray that starts ON the sphere, b*b - 4ac from terms of size |oc|^2 to r^2, so the rounding of the individual products decides
// whether a path re-hits the surface it leaves ("acne" at tMin) - and how often it does is part of the image the reference renders.  Two tiers:
//  - big_sphere_roots (the fast kernels, PRIM_SPHERE_BIG slots: r >> tMin, 03's ground): the coefficients in the reference's order, a = d.d of the kernel's own direction,
//    a correctly rounded root, the division by a as one fused correction.  The cancellation inside c dominates for a ray that leaves the sphere (fuzz scenes on the
//    ground sphere seen from other cameras: scene queries per sample off by up to -1.3 % with the fused form, within +-3e-4 with the reference's order), and a matters
//    for a ray that crosses it INSIDE (a rough metal ground scatters into the sphere): the far root is ~2r, and with a taken as 1 the hit point lies 2r(|d|^2 - 1) ~ 2e-5
//    off the surface - more than the reference's rounding noise there, so more false self-hits at the next bounce (+0.3 % scene queries; with a: see profiles/r03_path_fuzz.txt).
//  - strict_sphere_roots (reference arithmetic, RT_OPT_ARITHMETIC = RT_ARITH_REFERENCE: every sphere): a = d.d as the reference has it and true divisions.  Needed for
//    small spheres FAR from the ray's origin (|oc| >> r): |d|^2 is 1 to a few 1e-7 only and 4ac is of size |oc|^2, so a's last bits are as large as the noise.  It
//    only makes sense with the ray directions normalised as Ray's constructor does and the hit point rounded as the reference rounds it (REF in rt_path_device.h).
__device__ __forceinline__ void big_sphere_roots(f3 centre, float r2, f3 o, f3 d, float& t1, float& t2) {
    const f3 oc = F3(o.x - centre.x, o.y - centre.y, o.z - centre.z);
    const float a = (d.x * d.x + d.y * d.y) + d.z * d.z;
    const float bh = -((d.x * oc.x + d.y * oc.y) + d.z * oc.z);
    const float c = ((oc.x * oc.x + oc.y * oc.y) + oc.z * oc.z) - r2;
    const float s = sqrt_refined(bh * bh - a * c);      // NaN when the line misses: fails every compare
    // x / a for a = 1 + e, |e| < 1e-6: x - e*x in one rounding (e = a - 1 is exact; the e*e*x left out is 1e-12 x) - a division's result without a division's ten instructions
    const float e = a - 1.0f, x1 = bh - s, x2 = bh + s;
    t1 = fmaf(-e, x1, x1); t2 = fmaf(-e, x2, x2);
}
__device__ __forceinline__ void strict_sphere_roots(f3 centre, float r2, f3 o, f3 d, float& t1, float& t2) {
    const f3 oc = F3(o.x - centre.x, o.y - centre.y, o.z - centre.z);
    const float a = (d.x * d.x + d.y * d.y) + d.z * d.z;
    const float bh = -((d.x * oc.x + d.y * oc.y) + d.z * oc.z);
    const float c = ((oc.x * oc.x + oc.y * oc.y) + oc.z * oc.z) - r2;
    const float s = sqrt_refined(bh * bh - a * c);      // NaN when the line misses: fails every compare
    t1 = (bh - s) / a; t2 = (bh + s) / a;
}
// 1 / |v| as math::Normalize computes it (Common/Math.hpp:233-238: the length in the reference's operation order, then a division per component - here one reciprocal
// by true division; the per-component quotients v/len and v*(1/len) differ in the last bit at most)
__device__ __forceinline__ float strict_inv_length(f3 v) { return 1.0f / sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z); }
// origin + direction * t as the reference rounds it (product, then sum; Common/Ray.hpp, the hit records of the scene elements)
__device__ __forceinline__ f3 strict_point(f3 o, f3 d, float t) { return F3(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t); }
// Ray::intersectsPlane (Common/Ray.hpp:223-249).  Slot: G = (n, -), A = (p0, 0).
__device__ __forceinline__ bool strict_plane(f3 n, f3 p0, f3 o, f3 d, float tminp, float& t) {
    const float dn = (d.x * n.x) + (d.y * n.y) + (d.z * n.z);
    t = ((n.x * p0.x - n.x * o.x) + (n.y * p0.y - n.y * o.y) + (n.z * p0.z - n.z * o.z)) / dn;
    return (dn != 0.0f) & (t >= tminp);
}
// scene::Rectangle::intersectsRay (04 Scene/Rectangle.hpp:58-70) and one face of scene::Box::intersectsRay (Scene/Box.hpp:62-131):
// the ray goes to the object frame - origin R*(o - pos), direction Normalize(R*d) (Ray's constructor) with R = the three rows
// r0, r1, r2 of GetRotationMat(-orientation) - and meets the axis-aligned rectangle 'coordinate ax = off' there
// (Ray::intersectsAARectangleXy/Yz/Xz, Common/Ray.hpp:263-351): t = (off - o'[ax]) / d'[ax], then the two other coordinates at t
// against [ulo, uhi] x [vlo, vhi] (u, v = the remaining axes in increasing order).
// Slot: G = (r0, off), A = (pos, 0), G1 = (r1, ulo), G2 = (r2, vlo), G3 = (uhi, vhi, ax as int bits, -).
__device__ __forceinline__ bool strict_rect(f3 r0, f3 r1, f3 r2, f3 pos, int ax, float off, float ulo, float uhi, float vlo, float vhi,
                                            f3 o, f3 d, float tminp, float& t) {
    const f3 om = F3(o.x - pos.x, o.y - pos.y, o.z - pos.z);
    const f3 oo = F3(r0.x * om.x + r0.y * om.y + r0.z * om.z, r1.x * om.x + r1.y * om.y + r1.z * om.z, r2.x * om.x + r2.y * om.y + r2.z * om.z);
    f3 dd = F3(r0.x * d.x + r0.y * d.y + r0.z * d.z, r1.x * d.x + r1.y * d.y + r1.z * d.z, r2.x * d.x + r2.y * d.y + r2.z * d.z);
    const float len = sqrtf((dd.x * dd.x) + (dd.y * dd.y) + (dd.z * dd.z));      // math::Normalize (Math.hpp:233-238)
    dd = len > 0.0f ? F3(dd.x / len, dd.y / len, dd.z / len) : F3(0.0f, 0.0f, 0.0f);
    const float oa = ax == 0 ? oo.x : (ax == 1 ? oo.y : oo.z), da = ax == 0 ? dd.x : (ax == 1 ? dd.y : dd.z);
    const float ou = ax == 0 ? oo.y : oo.x, du = ax == 0 ? dd.y : dd.x;
    const float ov = ax == 2 ? oo.y : oo.z, dv = ax == 2 ? dd.y : dd.z;
    t = (off - oa) / da;
    const float u = ou + du * t, v = ov + dv * t;
    return (t >= tminp) & (u >= ulo) & (u <= uhi) & (v >= vlo) & (v <= vhi);
}
// Ray::intersectsTriangle (Common/Ray.hpp:150-212) with the normal and rows 0, 1 of Inverse([e2 e1 n]) precomputed on the host by
// the same float operations.  Slot: G = (n, -), A = (v0, 0), G1 = (row 0, -), G2 = (row 1, -).
__device__ __forceinline__ bool strict_tri(f3 n, f3 v0, f3 i0, f3 i1, f3 o, f3 d, float tminp, float& t) {
    const float dn = (d.x * n.x) + (d.y * n.y) + (d.z * n.z);
    t = ((n.x * v0.x - n.x * o.x) + (n.y * v0.y - n.y * o.y) + (n.z * v0.z - n.z * o.z)) / dn;
    const f3 ip = F3(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t);
    const f3 pt = F3(ip.x - v0.x, ip.y - v0.y, ip.z - v0.z);
    const float x = i0.x * pt.x + i0.y * pt.y + i0.z * pt.z, y = i1.x * pt.x + i1.y * pt.y + i1.z * pt.z;
    return (dn != 0.0f) & (t >= tminp) & (x >= 0.0f) & (y >= 0.0f) & (x + y <= 1.0f);
}
#pragma clang fp contract(fast)
// one strict slot at SceneMem address q of device type 'type' (wave-uniform)
template <bool LDS>
__device__ __forceinline__ bool strict_slot(const SceneMem<LDS> M, int type, uint32_t q, f3 o, f3 d, float tminp, float& t) {
    const float4 g = M.ld4(q), a = M.ld4(q + SLOT_A);
    if (type == PRIM_PLANE_STRICT) return strict_plane(F3(g), F3(a), o, d, tminp, t);
    const float4 g1 = M.ld4(q + SLOT_G1), g2 = M.ld4(q + SLOT_G2);
    if (type == PRIM_TRI_STRICT) return strict_tri(F3(g), F3(a), F3(g1), F3(g2), o, d, tminp, t);
    const float4 g3 = M.ld4(q + SLOT_G3);
    return strict_rect(F3(g), F3(g1), F3(g2), F3(a), __float_as_int(g3.z), g.w, g1.w, g3.x, g2.w, g3.y, o, d, tminp, t);
}

// Closest hit, 03 Scene::intersectsRay / 04 scene::List::intersectsRay (03 Utils.h:255-271, 04 Utils.h:362-390):
// tMax shrinks to the closest hit so far and a test accepts t <= tMax, so the LAST element wins exact ties.
// closest_hit_runs walks n_runs runs whose first slot is at SceneMem address q and updates (closest, best) in place, so a
// scene can be walked in pieces (k_path_tiled).
// All traced directions are unit vectors (Ray's constructor normalises), so a = d.d = 1 in the sphere test.
// PACKED: the run descriptors come from two 64-bit kernel-argument words (16 bits each: type | count << 3, at most 8 runs)
// instead of scalar loads: no memory latency per run, which matters where few waves are resident (k_whitted).
// EARLY (scenes of thousands of primitives, k_path_tiled): a wave skips the rest of a test that none of its lanes can pass -
// a sphere whose discriminant is negative in every lane (its roots would be NaN and fail every compare), a rectangle or
// triangle whose (padded) bounding sphere the ray's line misses in every lane.  Exact, and with small primitives the common case.
template <bool LDS, bool PACKED = false, bool EARLY = false, bool REF = RT_REF_RAYS>
__device__ __forceinline__ void closest_hit_runs(const SceneMem<LDS> M, const int4* __restrict__ runs, int n_runs, uint32_t q,
                                                 f3 o, f3 d, f3 inv_d, float tmin, float tminp, float& closest, uint32_t& best,
                                                 unsigned long long pack0 = 0ULL, unsigned long long pack1 = 0ULL) {
    // LDS addresses are VGPR operands of ds_read anyway: keep the cursor in a VGPR (one v_add per primitive) instead of
    // letting the compiler hold it in an SGPR and copy it out twice per primitive; global reads want it uniform (scalar loads)
    if (LDS) asm volatile("" : "+v"(q));
    int4 run = PACKED ? make_int4(0, 0, 0, 0) : runs[0];
    for (int r = 0; r < n_runs; ++r) {
        int type, count;
        if (PACKED) {
            const uint32_t desc = (uint32_t)((r < 4 ? pack0 : pack1) >> ((r & 3) * 16)) & 0xFFFFU;
            type = (int)(desc & 7U); count = (int)(desc >> 3);
        } else {
            type = run.x; count = run.z;
            run = runs[r + 1];     // prefetch (the section is padded by one record)
        }
        if (LDS) asm volatile("" : "+v"(q), "+v"(best), "+v"(closest));   // one definition point per run for the carried values
        if (is_sphere_type(type)) {
            const bool big = type == PRIM_SPHERE_BIG;      // wave-uniform (a run descriptor)
            // Ray::intersectsSphere (Common/Ray.hpp:94-137).  Roots t1 <= t2; "a root below tMin becomes tMax" (:127-128):
            // a ray that starts inside a sphere whose far root lies beyond tMax still reports a hit AT tMax (:129-134):
            // the element wins, the distance stays.  disc < 0 gives NaN roots, which fail every compare.
#define RT_SPHERE_DISC(G, B, DISC) f3 co_ = F3(G) - o; B = dot(d, co_); \
                DISC = fmaf(B, B, -fmaf(co_.z, co_.z, fmaf(co_.y, co_.y, fmaf(co_.x, co_.x, -(G).w))))
#define RT_SPHERE_TAKE(T1, T2, Q) { bool near_behind = (T1) < tmin; float tc = near_behind ? (T2) : (T1); bool ge = tc >= tmin, le = tc <= closest; \
                closest = (ge & le) ? tc : closest; best = (ge & (near_behind | le)) ? (Q) : best; }
#define RT_SPHERE_ROOTS(B, DISC, Q) { float s = sqrt_fast(DISC); float t1 = (B) - s, t2 = (B) + s; RT_SPHERE_TAKE(t1, t2, Q) }
            int i = 0;
            if (EARLY && !big) {   // four spheres per early-out decision: a quarter of the branches, independent discriminants overlap
                for (; i + 3 < count; i += 4, q += 4 * SLOT_BYTES) {
                    float4 g0 = M.ld4(q), g1 = M.ld4(q + SLOT_BYTES), g2 = M.ld4(q + 2 * SLOT_BYTES), g3 = M.ld4(q + 3 * SLOT_BYTES);
                    float b0, d0, b1, d1, b2, d2, b3, d3;
                    { RT_SPHERE_DISC(g0, b0, d0); }
                    { RT_SPHERE_DISC(g1, b1, d1); }
                    { RT_SPHERE_DISC(g2, b2, d2); }
                    { RT_SPHERE_DISC(g3, b3, d3); }
                    // fmaxf drops NaNs: all four negative (or NaN) <=> the maximum is negative (or NaN): one compare for the four
                    if (__builtin_amdgcn_ballot_w64(fmaxf(fmaxf(d0, d1), fmaxf(d2, d3)) >= 0.0f) == 0ULL) continue;
                    RT_SPHERE_ROOTS(b0, d0, q)
                    RT_SPHERE_ROOTS(b1, d1, q + SLOT_BYTES)
                    RT_SPHERE_ROOTS(b2, d2, q + 2 * SLOT_BYTES)
                    RT_SPHERE_ROOTS(b3, d3, q + 3 * SLOT_BYTES)
                }
            }
            for (; i < count; ++i, q += SLOT_BYTES) {
                float4 g = M.ld4(q);
                if (big) {
                    float t1, t2;
                    if (REF) strict_sphere_roots(F3(g), g.w, o, d, t1, t2); else big_sphere_roots(F3(g), g.w, o, d, t1, t2);
                    RT_SPHERE_TAKE(t1, t2, q)
                    continue;
                }
                float b, disc;
                { RT_SPHERE_DISC(g, b, disc); }
                if (EARLY && __builtin_amdgcn_ballot_w64(disc >= 0.0f) == 0ULL) continue;
                RT_SPHERE_ROOTS(b, disc, q)
            }
#undef RT_SPHERE_DISC
#undef RT_SPHERE_ROOTS
#undef RT_SPHERE_TAKE
        } else if (is_aaplane_type(type)) {
            // Ray::intersectsPlane (Ray.hpp:223-249) for a normal +-e_axis: n.p0 - n.o = +-(p0_a - o_a), d.n = +-d_a, so
            // t = (p0_a - o_a) * (1/d_a), 1/d shared by all such planes of the ray.  tminp = max(tMin, smallest positive
            // float) folds "t > 0 && t >= tMin" into one compare; d_a = 0 gives inf/NaN, which fails the range test.
#define RT_AA_LOOP(OA, IA) for (int i = 0; i < count; ++i, q += SLOT_BYTES) { \
                float t = (M.ld1(q) - (OA)) * (IA); bool hit = (t >= tminp) & (t <= closest); \
                closest = hit ? t : closest; best = hit ? q : best; }
            if (type == PRIM_AAPLANE_X) RT_AA_LOOP(o.x, inv_d.x)
            else if (type == PRIM_AAPLANE_Y) RT_AA_LOOP(o.y, inv_d.y)
            else RT_AA_LOOP(o.z, inv_d.z)
#undef RT_AA_LOOP
        } else if (type == PRIM_PLANE) {
            for (int i = 0; i < count; ++i, q += SLOT_BYTES) {
                float4 g = M.ld4(q);
                float t = (g.w - dot(F3(g), o)) * rcp_fast(dot(F3(g), d));
                bool hit = (t >= tminp) & (t <= closest);
                closest = hit ? t : closest; best = hit ? q : best;
            }
        } else if (is_strict_type(type)) {   // coplanar with a slot of another kind: the reference's own operation sequence (see PRIM_*_STRICT)
            const uint32_t step = type == PRIM_PLANE_STRICT ? SLOT_BYTES : 2 * SLOT_BYTES;
            const f3 ds = REF ? d : strict_dir(d);      // REF: the ray already holds the direction as the reference does
            for (int i = 0; i < count; ++i, q += step) {
                float t;
                const bool hit = strict_slot<LDS>(M, type, q, o, ds, tminp, t) & (t <= closest);
                closest = hit ? t : closest; best = hit ? q : best;
            }
        } else {
            // scene::Rectangle::intersectsRay (04 Scene/Rectangle.hpp:51-97, Ray.hpp:263-283): plane z' = 0 of the object
            // frame, then inclusive |x'| <= hw, |y'| <= hh.  Ray::intersectsTriangle (Ray.hpp:150-212): plane hit, then
            // coordinates in the [e2 e1 n] basis, inclusive x >= 0, y >= 0, x + y <= 1.  Both evaluate the in-plane
            // coordinates at the hit point p = o + t*d with the constant offset folded into the row's w.
            const bool rect = type == PRIM_RECT;
            for (int i = 0; i < count; ++i, q += 2 * SLOT_BYTES) {
                if (EARLY) {   // the line of the ray against the slot's bounding sphere; the slack covers the rounding of the discriminant (~1e-6 |co|^2)
                    float4 bs = M.ld4(q + SLOT_BOUND);
                    f3 co = F3(bs) - o;
                    float b = dot(d, co), cc = dot(co, co);
                    float disc = fmaf(b, b, bs.w - cc);
                    if (__builtin_amdgcn_ballot_w64(fmaf(cc, 4e-6f, disc) >= 0.0f) == 0ULL) continue;
                }
                float4 g = M.ld4(q), g1 = M.ld4(q + SLOT_G1), g2 = M.ld4(q + SLOT_G2);
                float t = (g.w - dot(F3(g), o)) * rcp_fast(dot(F3(g), d));
                f3 p = madd(d, t, o);
                float x = fmaf(g1.z, p.z, fmaf(g1.y, p.y, fmaf(g1.x, p.x, g1.w)));
                float y = fmaf(g2.z, p.z, fmaf(g2.y, p.y, fmaf(g2.x, p.x, g2.w)));
                bool in;
                if (rect) { float4 g3 = M.ld4(q + SLOT_G3); in = (fabsf(x) <= g3.x) & (fabsf(y) <= g3.y); }
                else      in = (x >= 0.0f) & (y >= 0.0f) & (x + y <= 1.0f);
                bool hit = (t >= tminp) & (t <= closest) & in;
                closest = hit ? t : closest; best = hit ? q : best;
            }
        }
    }
}
// The whole scene in one go: q = SceneMem address of the first slot.  Returns the address of the winning slot (0 = miss:
// address 0 is the runs section, or the first bytes of LDS, never a slot) and its distance.
template <bool LDS, bool EARLY = false, bool REF = RT_REF_RAYS>
__device__ __forceinline__ uint32_t closest_hit_slots(const SceneMem<LDS> M, const int4* __restrict__ runs, int n_runs, uint32_t q,
                                                      f3 o, f3 d, float tmin, float tminp, float tmax, float& t_out) {
    uint32_t best = 0U;
    float closest = tmax;
    closest_hit_runs<LDS, false, EARLY, REF>(M, runs, n_runs, q, o, d, F3(rcp_fast(d.x), rcp_fast(d.y), rcp_fast(d.z)), tmin, tminp, closest, best);
    t_out = closest;
    return best;
}

// Closest-hit bookkeeping on unsigned keys (the scene-specialised kernels, rt_jit.hip): if (k <= ck) { ck = k; best = K; }.
// Written as a subtraction whose borrow IS the compare (v_sub_co_u32, a full-rate VOP2, instead of a half-rate v_cmp) and two
// selects through VCC; the difference itself is not needed, the empty asm only keeps the compiler from folding the
// subtraction back into a compare.  "<=": among equal distances the LAST element wins, as in the reference's list walk.
__device__ __forceinline__ void key_take(uint32_t& ck, uint32_t& best, uint32_t k, uint32_t K) {
    uint32_t diff;
    const bool further = __builtin_usub_overflow(ck, k, &diff);      // k > ck
    asm volatile("" ::"v"(diff));
    ck = further ? ck : k;
    best = further ? best : K;
}

// The same bookkeeping with a plain unsigned compare (generation 3 of the generated query): v_cmp_le_u32 + two selects.
__device__ __forceinline__ void key_take3(uint32_t& ck, uint32_t& best, uint32_t k, uint32_t K) {
    const bool take = k <= ck;
    ck = take ? k : ck;
    best = take ? K : best;
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
