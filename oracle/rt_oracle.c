/*
 * rt_oracle.c - CPU restatement of the reference's per-pixel tracing path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the timed CPU baseline.  librt_mi355.so never links or calls it.
 *
 * Plain scalar C, recursive exactly where the reference is recursive, every function citing
 * the reference lines it restates (paths relative to /root/reference/Sources).  Arithmetic is
 * float with the reference's operator order; build with -ffp-contract=off so no FMA contraction
 * changes it.  The only deliberate difference: the reference draws random numbers from
 * wall-clock-seeded <random> engines (02_SoftShadows/Main.cpp:191-192, 03 Utils.h:21-23,
 * 04 Utils.h:18-19,135-136); here every draw is a pure function of
 * (seed, pixel, sample, counter) - the counter-based generator documented in DESIGN.md -
 * so that the HIP kernels can reproduce the very same paths.
 *
 * Pinning (see DESIGN.md "Oracle"): unit level against the reference's own header-only
 * code compiled from /root/reference (oracle/_ref, fixtures in tests/golden/kat_*.npz);
 * image level against the reference's recorded outputs the README_files PNGs and the ray counts
 * the survey measured on the reference.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stddef.h>
#include "../include/rt_mi355.h"

#define ORC_PI 3.14159265358979323846 /* Common/Math.hpp:11 (double literal) */

typedef struct { float x, y, z; } v3;

/* ---- Common/Math.hpp ------------------------------------------------------------------ */
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }          /* :89 */
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }          /* :94 */
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }                               /* :99 */
static inline v3 muls(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }            /* :74 */
static inline v3 divs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }            /* :79 */
static inline v3 mulv(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }         /* :84 */
static inline float length3(v3 v) { return sqrtf((v.x * v.x) + (v.y * v.y) + (v.z * v.z)); } /* :195 */
static inline v3 normalize3(v3 v) {                                                      /* :233-238 */
    float len = length3(v);
    if (len > 0) return V(v.x / len, v.y / len, v.z / len);
    return V(0, 0, 0);
}
static inline float dot3(v3 a, v3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); } /* :261 */
static inline v3 cross3(v3 a, v3 b) {                                                    /* :274-281 */
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 reflect3(v3 v, v3 n) { return sub(v, muls(muls(n, 2.0f), dot3(v, n))); } /* :291-294 */
static inline v3 refract3(v3 v, v3 n, float eta, int reflect_on_tir) {                   /* :306-312 */
    float d = dot3(v, n);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return reflect_on_tir ? reflect3(v, n) : V(0, 0, 0);
    return sub(muls(v, eta), muls(n, eta * d + sqrtf(k)));
}
static inline v3 mix3(v3 a, v3 b, float r) { return add(a, muls(sub(b, a), r)); }        /* :323-326 */
static inline float clampf(float v, float lo, float hi) { return (v < lo) ? lo : (hi < v) ? hi : v; } /* :1119 */

typedef struct { float m[9]; } m3; /* row-major storage, built from COLUMNS (Math.hpp:401-406) */
static inline m3 m3_cols(v3 i, v3 j, v3 k) {
    m3 r;
    r.m[0] = i.x; r.m[1] = j.x; r.m[2] = k.x;
    r.m[3] = i.y; r.m[4] = j.y; r.m[5] = k.y;
    r.m[6] = i.z; r.m[7] = j.z; r.m[8] = k.z;
    return r;
}
static inline v3 m3_mulv(const m3* a, v3 v) {                                            /* :425-432 */
    return V(a->m[0] * v.x + a->m[1] * v.y + a->m[2] * v.z,
             a->m[3] * v.x + a->m[4] * v.y + a->m[5] * v.z,
             a->m[6] * v.x + a->m[7] * v.y + a->m[8] * v.z);
}
static inline float m3_det(const m3* a) {                                                /* :583-591 */
    const float* m = a->m;
    return (m[0] * m[4] * m[8]) + (m[1] * m[5] * m[6]) + (m[2] * m[3] * m[7])
         - (m[2] * m[4] * m[6]) - (m[0] * m[5] * m[7]) - (m[1] * m[3] * m[8]);
}
static inline m3 m3_inverse(const m3* a) {                                               /* :635-654 */
    m3 r; memset(&r, 0, sizeof r);
    float det = m3_det(a);
    const float* m = a->m;
    if (det != 0) {
        float di = 1 / det;
        r.m[0] = (m[4] * m[8] - m[7] * m[5]) * di;
        r.m[1] = (m[2] * m[7] - m[1] * m[8]) * di;
        r.m[2] = (m[1] * m[5] - m[2] * m[4]) * di;
        r.m[3] = (m[5] * m[6] - m[3] * m[8]) * di;
        r.m[4] = (m[0] * m[8] - m[2] * m[6]) * di;
        r.m[5] = (m[3] * m[2] - m[0] * m[5]) * di;
        r.m[6] = (m[3] * m[7] - m[6] * m[4]) * di;
        r.m[7] = (m[6] * m[1] - m[0] * m[7]) * di;
        r.m[8] = (m[0] * m[4] - m[3] * m[1]) * di;
    }
    return r;
}
/* GetRotationMat, Euler degrees -> Mat3 (Math.hpp:913-930); the degree->radian product is
 * evaluated in double because M_PI is a double literal (:917-919). */
static m3 rotation_mat(v3 deg) {
    float ax = (float)(deg.x * (ORC_PI / 180.0f));
    float ay = (float)(deg.y * (ORC_PI / 180.0f));
    float az = (float)(deg.z * (ORC_PI / 180.0f));
    v3 i = V(cosf(az) * cosf(ay), sinf(az) * cosf(ay), -sinf(ay));
    v3 j = V(cosf(az) * sinf(ay) * sinf(ax) - sinf(az) * cosf(ax),
             sinf(az) * sinf(ay) * sinf(ax) + cosf(az) * cosf(ax), cosf(ay) * sinf(ax));
    v3 k = V(cosf(az) * sinf(ay) * cosf(ax) + sinf(az) * sinf(ax),
             sinf(az) * sinf(ay) * cosf(ax) - cosf(az) * sinf(ax), cosf(ay) * cosf(ax));
    return m3_cols(i, j, k);
}

/* ---- Common/Ray.hpp ------------------------------------------------------------------- */
typedef struct { v3 o, d; } ray_t;
static inline ray_t make_ray(v3 o, v3 d) { ray_t r; r.o = o; r.d = normalize3(d); return r; } /* :33-34 */

/* Ray::intersectsSphere (:94-137), including the "root < tMin becomes tMax" rule (:127-134). */
static int ray_sphere(const ray_t* r, v3 c, float radius, float tmin, float tmax, float* tout) {
    v3 oc = sub(r->o, c);
    float a = dot3(r->d, r->d);
    float b = 2.0f * dot3(r->d, oc);
    float cc = dot3(oc, oc) - (radius * radius);
    float disc = b * b - 4.0f * a * cc;
    if (disc < 0) return 0;
    float t1 = (-b - sqrtf(disc)) / (2.0f * a);
    float t2 = (-b + sqrtf(disc)) / (2.0f * a);
    if (t1 < tmin && t2 < tmin) return 0;
    if (t1 < tmin) t1 = tmax;
    if (t2 < tmin) t2 = tmax;
    float tr = (t2 < t1) ? t2 : t1; /* std::min(t1,t2) */
    if (tr <= tmax) { if (tout) *tout = tr; return 1; }
    return 0;
}
/* Ray::intersectsPlane (:223-249). */
static int ray_plane(const ray_t* r, v3 n, v3 p0, float tmin, float tmax, float* tout) {
    float d = dot3(r->d, n);
    if (d != 0) {
        float t = ((n.x * p0.x - n.x * r->o.x) + (n.y * p0.y - n.y * r->o.y) + (n.z * p0.z - n.z * r->o.z)) / d;
        if (t > 0 && t >= tmin && t <= tmax) { if (tout) *tout = t; return 1; }
    }
    return 0;
}
/* Ray::intersectsAARectangleXy (:263-283). */
static int ray_rect_xy(const ray_t* r, float z, float xmin, float xmax, float ymin, float ymax,
                       float tmin, float tmax, float* tout) {
    float t = (z - r->o.z) / r->d.z;
    if (t > 0 && t >= tmin && t <= tmax) {
        float x = r->o.x + r->d.x * t;
        float y = r->o.y + r->d.y * t;
        if (x >= xmin && x <= xmax && y >= ymin && y <= ymax) { if (tout) *tout = t; return 1; }
    }
    return 0;
}
/* Ray::intersectsAARectangleYz (:297-317) and ...Xz (:331-351). */
static int ray_rect_yz(const ray_t* r, float x, float ymin, float ymax, float zmin, float zmax, float tmin, float tmax, float* tout) {
    float t = (x - r->o.x) / r->d.x;
    if (t > 0 && t >= tmin && t <= tmax) {
        float y = r->o.y + r->d.y * t;
        float z = r->o.z + r->d.z * t;
        if (y >= ymin && y <= ymax && z >= zmin && z <= zmax) { if (tout) *tout = t; return 1; }
    }
    return 0;
}
static int ray_rect_xz(const ray_t* r, float y, float xmin, float xmax, float zmin, float zmax, float tmin, float tmax, float* tout) {
    float t = (y - r->o.y) / r->d.y;
    if (t > 0 && t >= tmin && t <= tmax) {
        float x = r->o.x + r->d.x * t;
        float z = r->o.z + r->d.z * t;
        if (x >= xmin && x <= xmax && z >= zmin && z <= zmax) { if (tout) *tout = t; return 1; }
    }
    return 0;
}
/* Ray::intersectsTriangle (:150-212): plane hit, then barycentrics through Inverse([e2 e1 n]). */
static int ray_triangle(const ray_t* r, v3 v0, v3 v1, v3 v2, float tmin, float tmax, float* tout, float* bary) {
    v3 e1 = sub(v1, v0), e2 = sub(v2, v0);
    v3 n = normalize3(cross3(e2, e1));
    float d = dot3(r->d, n);
    if (d != 0) {
        float t = ((n.x * v0.x - n.x * r->o.x) + (n.y * v0.y - n.y * r->o.y) + (n.z * v0.z - n.z * r->o.z)) / d;
        if (t > 0 && t >= tmin && t <= tmax) {
            v3 ip = add(r->o, muls(r->d, t));
            v3 pt = sub(ip, v0);
            m3 sp = m3_cols(e2, e1, n);
            m3 inv = m3_inverse(&sp);
            v3 res = m3_mulv(&inv, pt);
            if (res.x >= 0.0f && res.y >= 0.0f && res.x + res.y <= 1.0f) {
                if (bary) { bary[0] = res.x; bary[1] = res.y; }
                if (tout) *tout = t;
                return 1;
            }
        }
    }
    return 0;
}

/* ---- counter-based RNG (DESIGN.md "RNG"; replaces the reference's <random> engines) ---- */
static inline uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
static inline uint32_t rng_pixel_key(uint32_t seed, uint32_t pixel) {
    return lowbias32(pixel + 0x9E3779B9U * seed + 0x85EBCA6BU);
}
static inline uint32_t rng_sample_key(uint32_t pixel_key, uint32_t sample) {
    return lowbias32(pixel_key ^ (sample * 0x9E3779B1U));
}
static inline uint32_t rng_draw(uint32_t sample_key, uint32_t ctr) {
    return lowbias32(sample_key + (ctr + 1U) * 0x9E3779B9U);
}
static inline float rng_u01(uint32_t sample_key, uint32_t ctr) { /* uniform in [0,1), 24 bits */
    return (float)(rng_draw(sample_key, ctr) >> 8) * (1.0f / 16777216.0f);
}
/* two 16-bit uniforms from ONE draw: a = low half, b = high half, each k/65536 */
static inline void rng_pair(uint32_t sample_key, uint32_t ctr, float* a, float* b) {
    uint32_t r = rng_draw(sample_key, ctr);
    *a = (float)(r & 0xFFFFU) * (1.0f / 65536.0f);
    *b = (float)(r >> 16) * (1.0f / 65536.0f);
}
uint32_t orc_rng_draw(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t ctr) {
    return rng_draw(rng_sample_key(rng_pixel_key(seed, pixel), sample), ctr);
}

/* ---- scene queries --------------------------------------------------------------------- */
typedef struct {
    v3 point, normal;
    float t;
    int front;
    int material; /* index, -1 = none */
} hit_t;          /* 03 Utils.h:98-110, 04 Utils.h:171-183 */

static inline v3 pv(const float* f) { return V(f[0], f[1], f[2]); }

/* 01/02 SceneElement::intersectsRay (Sphere.hpp:43-52, Plane.hpp:43-52): t and the raw normal
 * (NOT face-forwarded).  Triangle follows the Plane wrapper with Ray::intersectsTriangle. */
static int prim_hit_whitted(const rt_prim* p, const ray_t* r, float tmin, float tmax, float* tout, v3* nout) {
    float t = 0;
    switch (p->type) {
    case RT_PRIM_SPHERE:
        if (ray_sphere(r, pv(p->f), p->f[3], tmin, tmax, &t)) {
            *tout = t;
            *nout = normalize3(sub(add(r->o, muls(r->d, t)), pv(p->f)));
            return 1;
        }
        return 0;
    case RT_PRIM_PLANE:
        if (ray_plane(r, pv(p->f + 3), pv(p->f), tmin, tmax, &t)) { *tout = t; *nout = pv(p->f + 3); return 1; }
        return 0;
    case RT_PRIM_TRIANGLE:
        if (ray_triangle(r, pv(p->f), pv(p->f + 3), pv(p->f + 6), tmin, tmax, &t, NULL)) {
            *tout = t;
            *nout = normalize3(cross3(sub(pv(p->f + 6), pv(p->f)), sub(pv(p->f + 3), pv(p->f))));
            return 1;
        }
        return 0;
    default:
        return 0;
    }
}

/* 03 Sphere::intersectsRay (Sphere.hpp:46-78); 04 scene::Sphere (Scene/Sphere.hpp:47-79),
 * scene::Plane (Scene/Plane.hpp:44-71), scene::Rectangle (Scene/Rectangle.hpp:51-97). */
static int prim_hit_path(const rt_prim* p, const ray_t* r, float tmin, float tmax, hit_t* h) {
    float t = 0;
    switch (p->type) {
    case RT_PRIM_SPHERE:
        if (ray_sphere(r, pv(p->f), p->f[3], tmin, tmax, &t)) {
            h->t = t;
            h->point = add(r->o, muls(r->d, t));
            h->normal = normalize3(sub(h->point, pv(p->f)));
            h->front = 1;
            h->material = p->material;
            if (p->flags & RT_PRIM_FLAG_FLIP_NORMALS) h->normal = neg(h->normal);
            if (dot3(neg(r->d), h->normal) < 0.0f) { h->normal = neg(h->normal); h->front = 0; }
            return 1;
        }
        return 0;
    case RT_PRIM_PLANE:
        if (ray_plane(r, pv(p->f + 3), pv(p->f), tmin, tmax, &t)) {
            h->t = t;
            h->point = add(r->o, muls(r->d, t));
            h->normal = pv(p->f + 3);
            h->front = 1;
            h->material = p->material;
            if (dot3(neg(r->d), h->normal) < 0.0f) { h->normal = neg(h->normal); h->front = 0; }
            return 1;
        }
        return 0;
    case RT_PRIM_RECT: {
        v3 orient = pv(p->f + 5);
        m3 inv = rotation_mat(neg(orient));                                   /* :58 */
        v3 pos = pv(p->f);
        ray_t tr = make_ray(m3_mulv(&inv, sub(r->o, pos)), m3_mulv(&inv, r->d)); /* :61-63 */
        float hw = p->f[3] / 2.0f, hh = p->f[4] / 2.0f;
        if (ray_rect_xy(&tr, 0.0f, -hw, hw, -hh, hh, tmin, tmax, &t)) {
            h->t = t;
            h->point = add(r->o, muls(r->d, t));
            h->normal = V(0.0f, 0.0f, 1.0f);
            h->front = 1;
            h->material = p->material;
            if (dot3(neg(tr.d), h->normal) < 0.0f) { h->normal = neg(h->normal); h->front = 0; }
            m3 rot = rotation_mat(orient);
            h->normal = m3_mulv(&rot, h->normal);                               /* :92 */
            return 1;
        }
        return 0;
    }
    case RT_PRIM_BOX: {   /* scene::Box::intersectsRay (04 Scene/Box.hpp:55-158), face by face, quirks kept */
        v3 orient = pv(p->f + 6);
        m3 inv = rotation_mat(neg(orient));                                   /* :62 */
        v3 pos = pv(p->f);
        ray_t tr = make_ray(m3_mulv(&inv, sub(r->o, pos)), m3_mulv(&inv, r->d)); /* :65-67 */
        float hw = p->f[3] / 2.0f, hh = p->f[4] / 2.0f, hl = p->f[5] / 2.0f;   /* :70-72 */
        float tclosest = tmax;
        int any = 0;
        v3 nc = V(0, 0, 0);
        /* the two Z faces take y in [-halfHeight, +halfLength] (:88, :95), as written in the reference */
        if (ray_rect_xy(&tr, -hl, -hw, hw, -hh, hl, tmin, tclosest, &t)) { any = 1; tclosest = t; nc = V(0.0f, 0.0f, -1.0f); }
        if (ray_rect_xy(&tr, hl, -hw, hw, -hh, hl, tmin, tclosest, &t)) { any = 1; tclosest = t; nc = V(0.0f, 0.0f, 1.0f); }
        if (ray_rect_yz(&tr, -hw, -hh, hh, -hl, hl, tmin, tclosest, &t)) { any = 1; tclosest = t; nc = V(-1.0f, 0.0f, 0.0f); }
        if (ray_rect_yz(&tr, hw, -hh, hh, -hl, hl, tmin, tclosest, &t)) { any = 1; tclosest = t; nc = V(1.0f, 0.0f, 0.0f); }
        if (ray_rect_xz(&tr, -hh, -hw, hw, -hl, hl, tmin, tclosest, &t)) { any = 1; tclosest = t; nc = V(0.0f, -1.0f, 0.0f); }
        if (ray_rect_xz(&tr, hh, -hw, hw, -hl, hl, tmin, tclosest, &t)) { any = 1; tclosest = t; nc = V(0.0f, 1.0f, 0.0f); }
        if (any) {
            h->t = t;                                                           /* :136: the last accepted face is the closest */
            h->point = add(r->o, muls(r->d, t));
            h->normal = nc;
            h->front = 1;
            h->material = p->material;
            if (p->flags & RT_PRIM_FLAG_FLIP_NORMALS) h->normal = neg(h->normal);
            if (dot3(neg(tr.d), h->normal) < 0.0f) { h->normal = neg(h->normal); h->front = 0; }
            m3 rot = rotation_mat(orient);
            h->normal = m3_mulv(&rot, h->normal);                               /* :154 */
        }
        return any;
    }
    case RT_PRIM_TRIANGLE: /* extension: Plane wrapper semantics over Ray::intersectsTriangle */
        if (ray_triangle(r, pv(p->f), pv(p->f + 3), pv(p->f + 6), tmin, tmax, &t, NULL)) {
            h->t = t;
            h->point = add(r->o, muls(r->d, t));
            h->normal = normalize3(cross3(sub(pv(p->f + 6), pv(p->f)), sub(pv(p->f + 3), pv(p->f))));
            h->front = 1;
            h->material = p->material;
            if (dot3(neg(r->d), h->normal) < 0.0f) { h->normal = neg(h->normal); h->front = 0; }
            return 1;
        }
        return 0;
    default:
        return 0;
    }
}

/* 03 Scene::intersectsRay (Utils.h:255-271) == 04 scene::List::intersectsRay (Utils.h:362-390):
 * shrinking tMax, accept on t <= tMax, so the LAST element wins exact ties. */
static int scene_hit_path(const rt_scene* s, const ray_t* r, float tmin, float tmax, hit_t* out) {
    hit_t h; memset(&h, 0, sizeof h);
    int any = 0;
    float closest = tmax;
    for (int i = 0; i < s->n_prims; i++) {
        if (prim_hit_path(&s->prims[i], r, tmin, closest, &h)) { any = 1; closest = h.t; *out = h; }
    }
    return any;
}

typedef struct { uint64_t rays, shadow_rays; } counters_t;

/* ---- 01_Basic / 02_SoftShadows ---------------------------------------------------------- */
typedef struct {
    const rt_scene* scene;
    int max_depth, shadow_samples;
    uint32_t sample_key;
    counters_t* cnt;
} wctx_t;

/* RandomVectorToLightSphere (02 Main.cpp:539-566); vBias x,y,z in U(-1,1), rBias in U(0,1): two pair draws. */
static v3 random_vector_to_light(const wctx_t* w, v3 p, const rt_light* l, uint32_t ctr_base) {
    v3 lp = pv(l->position);
    v3 to = normalize3(sub(lp, p));
    /* uniform_real_distribution(a,b): (b-a)*u + a */
    float u0, u1, u2, rb;
    rng_pair(w->sample_key, ctr_base + 0, &u0, &u1);
    rng_pair(w->sample_key, ctr_base + 2, &u2, &rb);
    v3 vb = V(2.0f * u0 + -1.0f, 2.0f * u1 + -1.0f, 2.0f * u2 + -1.0f);
    v3 pl = normalize3(cross3(to, add(to, vb)));
    v3 target = add(lp, muls(muls(pl, l->radius), rb));
    return normalize3(sub(target, p));
}

/* TraceTay of 01 (Main.cpp:372-496) and 02 (Main.cpp:389-531).  color==NULL is the shadow-ray
 * form.  'node' numbers the recursion tree (root 1, reflect child 2n, refract child 2n+1) and
 * only feeds the RNG counter of 02. */
static int trace_whitted(const wctx_t* w, const ray_t* ray, int with_lights, float tmin, float tmax,
                         v3* color, int depth, uint32_t node) {
    const rt_scene* s = w->scene;
    int hit = 0;
    float nearest = 3.402823466e+38F; /* numeric_limits<float>::max() */
    v3 nrm = V(0, 0, 0);
    int idx = 0;
    if (depth <= w->max_depth) {                                                        /* :386 */
        if (color) w->cnt->rays++; else w->cnt->shadow_rays++;
        for (int i = 0; i < s->n_prims; i++) {
            float t; v3 n;
            if (prim_hit_whitted(&s->prims[i], ray, tmin, tmax, &t, &n) && t < nearest) { /* :393 */
                hit = 1; nearest = t; nrm = n; idx = i;
            }
        }
    }
    if (hit && color) {
        static const rt_material none = {RT_MAT_PHONG, {1, 1, 1}, 0.0f, 16.0f, 1.0f, 1.0f, 1.0f, 0, 0, 0}; /* Types.h:51-58 */
        int mi = s->prims[idx].material;
        const rt_material* m = (mi >= 0 && mi < s->n_materials) ? &s->materials[mi] : &none;
        v3 p = add(ray->o, muls(ray->d, nearest));                                      /* :414 */
        v3 primary = V(0, 0, 0), secondary = V(0, 0, 0);
        if (m->primary_to_secondary > 0.0f) {                                           /* :421 */
            v3 diffuse = V(0, 0, 0), specular = V(0, 0, 0);
            int nl = with_lights ? s->n_lights : 0;
            for (int li = 0; li < nl; li++) {
                const rt_light* l = &s->lights[li];
                v3 to = sub(pv(l->position), p);
                v3 tod = normalize3(to);
                float intensity = 1.0f;
                if (w->shadow_samples <= 0) {                                           /* 01 :435-436 */
                    ray_t sr = make_ray(p, to);
                    if (trace_whitted(w, &sr, 0, 0.01f, length3(to), NULL, depth + 1, 0)) continue;
                } else {                                                                /* 02 :452-467 */
                    float per = 1.0f / (float)w->shadow_samples;
                    for (int i = 0; i < w->shadow_samples; i++) {
                        uint32_t ctr = ((node * (uint32_t)s->n_lights + (uint32_t)li) * (uint32_t)w->shadow_samples + (uint32_t)i) * 4U;
                        v3 rd = random_vector_to_light(w, p, l, ctr);
                        ray_t sr = make_ray(p, rd);
                        if (trace_whitted(w, &sr, 0, 0.01f, length3(to) - l->radius, NULL, depth + 1, 0)) intensity -= per;
                    }
                }
                float dd = dot3(tod, nrm); if (!(0.0f < dd)) dd = 0.0f;      /* std::max(0.0f, x) */
                float sd = dot3(reflect3(neg(tod), nrm), neg(ray->d)); if (!(0.0f < sd)) sd = 0.0f;
                float sp = powf(sd, m->shininess);
                if (w->shadow_samples <= 0) {                                           /* 01 :439-442 */
                    diffuse = add(diffuse, muls(pv(l->color), dd));
                    specular = add(specular, muls(pv(l->color), sp));
                } else {                                                                /* 02 :474-477 */
                    diffuse = add(diffuse, muls(muls(pv(l->color), dd), intensity));
                    specular = add(specular, muls(muls(pv(l->color), sp), intensity));
                }
            }
            primary = add(mulv(diffuse, pv(m->albedo)), muls(specular, m->specular_intensity)); /* :446 */
        }
        if (m->primary_to_secondary < 1.0f) {                                           /* :450 */
            v3 refl = V(0, 0, 0), refr = V(0, 0, 0);
            if (m->reflect_to_refract > 0.0f) {                                         /* :457-461 */
                ray_t rr = make_ray(p, reflect3(ray->d, nrm));
                trace_whitted(w, &rr, with_lights, 0.001f, tmax, &refl, depth + 1, node * 2U);
            }
            if (m->reflect_to_refract < 1.0f) {                                         /* :464-477 */
                float eta = m->refraction_eta;
                if (dot3(nrm, neg(ray->d)) < 0.0f) { nrm = neg(nrm); eta = 1.0f / eta; }
                ray_t rr = make_ray(p, refract3(ray->d, nrm, eta, 1));
                trace_whitted(w, &rr, with_lights, 0.001f, tmax, &refr, depth + 1, node * 2U + 1U);
            }
            float wr = 1.0f - m->reflect_to_refract; if (!(0.0f < wr)) wr = 0.0f;       /* std::max(x, 0.0f) */
            secondary = mix3(refl, refr, wr);                                           /* :480 */
        }
        float wp = 1.0f - m->primary_to_secondary; if (!(0.0f < wp)) wp = 0.0f;
        *color = mix3(primary, secondary, wp);                                          /* :484 */
    } else if (color) {
        *color = V(0.2f, 0.7f, 0.8f);                                                   /* :489-493 */
    }
    return hit;
}

static inline uint8_t quant(float c) {                                  /* static_cast<BYTE>(Clamp(c,0,1)*255.0f) */
    float v = clampf(c, 0.0f, 1.0f) * 255.0f;
    if (!(v >= 0.0f)) return 0; /* NaN: undefined in the reference */
    return (uint8_t)v;
}

/* Render of 01 (Main.cpp:324-359) / 02 (Main.cpp:341-376) for rows [y0,y1).
 * rgb: W*H*3 floats (may be NULL), bgra: W*H*4 bytes (may be NULL), counts[2] += rays, shadow rays. */
int orc_render_whitted(const rt_scene* scene, const rt_camera* cam, const rt_whitted_params* prm,
                       int width, int height, int y0, int y1, float* rgb, uint8_t* bgra, uint64_t* counts) {
    if (!scene || !cam || !prm || width <= 0 || height <= 0) return RT_ERR_BAD_ARG;
    float w = (float)width, h = (float)height;
    float fov_rad = (float)(cam->fov_deg / (180.0f / ORC_PI));                          /* :333 */
    counters_t cnt = {0, 0};
    wctx_t wc; wc.scene = scene; wc.max_depth = prm->max_depth; wc.shadow_samples = prm->shadow_samples; wc.cnt = &cnt;
    for (int j = y0; j < y1; j++) {
        for (int i = 0; i < width; i++) {
            float x = (2.0f * ((float)i + 0.5f) / w - 1.0f) * tanf(fov_rad / 2.0f) * w / h;  /* :340 */
            float y = -(2.0f * ((float)j + 0.5f) / h - 1.0f) * tanf(fov_rad / 2.0f);         /* :341 */
            ray_t ray = make_ray(V(0, 0, 0), V(x, y, -1.0f));
            uint32_t pixel = (uint32_t)j * (uint32_t)width + (uint32_t)i;
            wc.sample_key = rng_sample_key(rng_pixel_key(prm->seed, pixel), 0);
            v3 c = V(0, 0, 0);
            trace_whitted(&wc, &ray, 1, 0.0f, 1000.0f, &c, 0, 1);                         /* :348 */
            size_t o = (size_t)pixel;
            if (rgb) { rgb[o * 3 + 0] = c.x; rgb[o * 3 + 1] = c.y; rgb[o * 3 + 2] = c.z; }
            if (bgra) { bgra[o * 4 + 0] = quant(c.z); bgra[o * 4 + 1] = quant(c.y); bgra[o * 4 + 2] = quant(c.x); bgra[o * 4 + 3] = 255; }
        }
    }
    if (counts) { counts[0] += cnt.rays; counts[1] += cnt.shadow_rays; }
    return RT_OK;
}

/* ---- 03_PathTracingBasics / 04_PathTracingLights ---------------------------------------- */
typedef struct {
    const rt_scene* scene;
    int variant, max_depth;
    float tmin, tmax;
    uint32_t sample_key;
    counters_t* cnt;
} pctx_t;

/* RndHemisphereVec of 03 (Utils.h:64-88): fi = U(0,1)*360/57.2958, theta = U(-1,1)*thetaMax/57.2958. */
static v3 rnd_hemisphere_03(v3 dir, float theta_max, float u_fi, float u_theta) {
    v3 b = normalize3(cross3(dir, add(dir, V(0.01f, 0.01f, 0.01f))));
    v3 c = normalize3(cross3(dir, b));
    float fi = (u_fi * 360.0f) / 57.2958f;
    float theta = ((2.0f * u_theta + -1.0f) * theta_max) / 57.2958f;
    v3 d = add(muls(b, cosf(fi)), muls(c, sinf(fi)));
    return add(muls(dir, cosf(theta)), muls(d, sinf(theta)));
}
/* RndHemisphereVec3 of 04 (Utils.h:132-157): theta = acos(U(cos(thetaMax/57.2958), 1)). */
static v3 rnd_hemisphere_04(v3 dir, float theta_max, float u_fi, float u_h) {
    float hmin = cosf(theta_max / 57.2958f);
    v3 b = normalize3(cross3(dir, add(dir, V(0.01f, 0.01f, 0.01f))));
    v3 c = normalize3(cross3(dir, b));
    float fi = (u_fi * 360.0f) / 57.2958f;
    float theta = acosf((1.0f - hmin) * u_h + hmin);
    v3 d = add(muls(b, cosf(fi)), muls(c, sinf(fi)));
    return add(muls(dir, cosf(theta)), muls(d, sinf(theta)));
}
void orc_rnd_hemisphere(int variant, const float* dir, float theta_max, float u1, float u2, float* out) {
    v3 r = variant == RT_PATH_SKY ? rnd_hemisphere_03(pv(dir), theta_max, u1, u2) : rnd_hemisphere_04(pv(dir), theta_max, u1, u2);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* Schlick reflectance (03 Materials.hpp:188-195 == 04 Materials/Refractive.hpp:20-27). */
static float reflectance(v3 v, v3 n, float ratio) {
    float cosine = dot3(neg(v), n); if (!(0.0f < cosine)) cosine = 0.0f; /* std::max(x,0.0f) */
    float r0 = (1.0f - ratio) / (1.0f + ratio);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powf((1.0f - cosine), 5);
}

/* Material::scatter of 03 (Materials.hpp:50-75,142-169,233-265) and materials::*::scatteredRay of
 * 04 (Diffuse.hpp:64-82, Metal.hpp:67-88, Refractive.hpp:80-104).  Returns the scattered ray and
 * attenuation.  RNG: ONE draw per bounce, counter 4(k+1) at recursion depth k: its 16-bit halves are the pair (fi, theta/height)
 * of the hemisphere/cone samplers, its top 24 bits the dielectric's reflect-or-refract threshold (only one of the two is used). */
static ray_t scatter(const pctx_t* pc, const rt_material* m, const ray_t* in, const hit_t* h, int depth, v3* att) {
    uint32_t c0 = 4U * ((uint32_t)depth + 1U);
    float ua, ub;
    rng_pair(pc->sample_key, c0, &ua, &ub);   /* (phi, theta/height) */
    v3 dir;
    switch (m->type) {
    case RT_MAT_DIFFUSE:
        if (pc->variant == RT_PATH_SKY) {
            dir = rnd_hemisphere_03(h->normal, 90.0f, ua, ub);
            *att = pv(m->albedo);
            return make_ray(h->point, dir);
        } else {
            dir = rnd_hemisphere_04(h->normal, 90.0f, ua, ub);
            ray_t r = make_ray(h->point, dir);
            float a = dot3(r.d, h->normal); if (!(0.0f < a)) a = 0.0f;
            *att = muls(pv(m->albedo), a);
            return r;
        }
    case RT_MAT_METAL:
        dir = reflect3(in->d, h->normal);
        if (pc->variant == RT_PATH_SKY) {
            if (m->roughness > 0.0f)
                dir = rnd_hemisphere_03(dir, 60.0f * m->roughness, ua, ub);
            *att = pv(m->albedo);
            return make_ray(h->point, dir);
        } else {
            dir = rnd_hemisphere_04(dir, 60.0f * m->roughness, ua, ub);
            ray_t r = make_ray(h->point, dir);
            float a = dot3(r.d, h->normal); if (!(0.0f < a)) a = 0.0f;
            *att = muls(pv(m->albedo), a);
            return r;
        }
    case RT_MAT_DIELECTRIC: {
        float idx = h->front ? m->refraction_index : 1.0f / m->refraction_index;
        dir = reflect3(in->d, h->normal);
        if (reflectance(in->d, h->normal, 1.0f / idx) < rng_u01(pc->sample_key, c0))
            dir = refract3(in->d, h->normal, idx, 1);
        *att = V(1.0f, 1.0f, 1.0f);
        return make_ray(h->point, dir);
    }
    default:
        *att = V(0, 0, 0);
        return make_ray(V(0, 0, 0), V(0, 0, -1.0f)); /* Ray() (Ray.hpp:25) */
    }
}

/* TraceTay of 03 (Main.cpp:380-435) and 04 (Main.cpp:431-502). */
static int trace_path(const pctx_t* pc, const ray_t* ray, v3* color, int depth) {
    if (depth > pc->max_depth) { *color = V(0, 0, 0); return 0; }
    hit_t h; memset(&h, 0, sizeof h);
    pc->cnt->rays++;
    if (scene_hit_path(pc->scene, ray, pc->tmin, pc->tmax, &h)) {
        if (h.material >= 0 && h.material < pc->scene->n_materials) {
            const rt_material* m = &pc->scene->materials[h.material];
            v3 result = V(0, 0, 0);
            int scatters = (m->type == RT_MAT_DIFFUSE || m->type == RT_MAT_METAL || m->type == RT_MAT_DIELECTRIC);
            if (scatters) {
                v3 att = V(0, 0, 0), child = V(0, 0, 0);
                ray_t sr = scatter(pc, m, ray, &h, depth, &att);
                trace_path(pc, &sr, &child, depth + 1);
                result = add(result, mulv(att, child));
                result = divs(result, 1.0f); /* SAMPLES_PER_RAY 1 */
            }
            if (pc->variant == RT_PATH_LIGHTS && m->type == RT_MAT_LIGHT && h.front)     /* Light.hpp:47-54 */
                result = add(result, pv(m->albedo));
            *color = result;
        } else {
            *color = V(0, 0, 0);
        }
        return 1;
    }
    if (pc->variant == RT_PATH_SKY) {                                                    /* 03 :430-432 */
        double hh = 0.5 * (ray->d.y + 1.0);
        *color = mix3(V(1.0f, 1.0f, 1.0f), V(0.5f, 0.7f, 1.0f), (float)hh);
    } else {
        *color = V(0, 0, 0);                                                             /* 04 :498 */
    }
    return 0;
}

/* Render of 03 (Main.cpp:310-370) / 04 (Main.cpp:324-397) for rows [y0,y1) and samples
 * [sample_begin,sample_end).  sum: W*H*3 float per-pixel SUMS over the rendered samples (+=
 * semantics are NOT used: overwritten), sumsq (optional): per-pixel sum of squared sample
 * luminance-free per-channel values (for variance estimates), bgra (optional, only meaningful
 * when the full sample range was rendered): gamma-2 + truncation as the reference. */
static int render_path_rows(const rt_scene* scene, const rt_camera* cam, const rt_path_params* prm,
                            int width, int height, int y0, int y1, float* sum, float* sumsq, uint8_t* bgra, uint64_t* counts, uint32_t* px_rays) {
    if (!scene || !cam || !prm || width <= 0 || height <= 0) return RT_ERR_BAD_ARG;
    float w = (float)width, h = (float)height;
    float fov_rad = (float)(cam->fov_deg / (180.0f / ORC_PI));
    int s0 = prm->sample_begin, s1 = prm->sample_end;
    if (s0 == 0 && s1 == 0) s1 = prm->spp;
    counters_t cnt = {0, 0};
    pctx_t pc; pc.scene = scene; pc.variant = prm->variant; pc.max_depth = prm->max_depth;
    pc.tmin = prm->t_min; pc.tmax = prm->t_max; pc.cnt = &cnt;
    v3 cpos = pv(cam->position), corient = pv(cam->orientation_deg);
    for (int j = y0; j < y1; j++) {
        for (int i = 0; i < width; i++) {
            uint32_t pixel = (uint32_t)j * (uint32_t)width + (uint32_t)i;
            uint32_t pkey = rng_pixel_key(prm->seed, pixel);
            v3 pix = V(0, 0, 0), pix2 = V(0, 0, 0);
            const uint64_t rays_before = cnt.rays;
            for (int s = s0; s < s1; s++) {
                pc.sample_key = rng_sample_key(pkey, (uint32_t)s);
                float bx = 0.5f, by = 0.5f;
                if (prm->spp > 1) rng_pair(pc.sample_key, 0, &bx, &by);                       /* :331 */
                float x = (2.0f * ((float)i + bx) / w - 1.0f) * tanf(fov_rad / 2.0f) * w / h;
                float y = -(2.0f * ((float)j + by) / h - 1.0f) * tanf(fov_rad / 2.0f);
                m3 rot = rotation_mat(corient);                                           /* :338 */
                ray_t ray = make_ray(cpos, m3_mulv(&rot, V(x, y, -1.0f)));
                v3 c = V(0, 0, 0);
                trace_path(&pc, &ray, &c, 0);
                pix = add(pix, c);
                pix2 = add(pix2, mulv(c, c));
            }
            size_t o = (size_t)pixel;
            if (px_rays) px_rays[o] = (uint32_t)(cnt.rays - rays_before);
            if (sum) { sum[o * 3 + 0] = pix.x; sum[o * 3 + 1] = pix.y; sum[o * 3 + 2] = pix.z; }
            if (sumsq) { sumsq[o * 3 + 0] = pix2.x; sumsq[o * 3 + 1] = pix2.y; sumsq[o * 3 + 2] = pix2.z; }
            if (bgra) {
                v3 m = divs(pix, (float)prm->spp);                                        /* :352 */
                m = V(sqrtf(m.x), sqrtf(m.y), sqrtf(m.z));                                /* :355-359 */
                bgra[o * 4 + 0] = quant(m.z); bgra[o * 4 + 1] = quant(m.y); bgra[o * 4 + 2] = quant(m.x); bgra[o * 4 + 3] = 255;
            }
        }
    }
    if (counts) { counts[0] += cnt.rays; counts[1] += cnt.shadow_rays; }
    return RT_OK;
}

int orc_render_path(const rt_scene* scene, const rt_camera* cam, const rt_path_params* prm,
                    int width, int height, int y0, int y1, float* sum, float* sumsq, uint8_t* bgra, uint64_t* counts) {
    return render_path_rows(scene, cam, prm, width, height, y0, y1, sum, sumsq, bgra, counts, NULL);
}
/* The same, also reporting the number of scene queries each pixel made (debugging aid for parity tests). */
int orc_render_path_px(const rt_scene* scene, const rt_camera* cam, const rt_path_params* prm,
                       int width, int height, int y0, int y1, float* sum, uint64_t* counts, uint32_t* px_rays) {
    return render_path_rows(scene, cam, prm, width, height, y0, y1, sum, NULL, NULL, counts, px_rays);
}

/* The tail of Render (03 Main.cpp:352-367): sum/divisor, optional sqrt gamma, Clamp, truncation, B,G,R,255
 * (used to check rt_resolve_device). */
void orc_resolve(const float* rgb, int n_pixels, float divisor, int gamma, uint8_t* bgra) {
    for (int p = 0; p < n_pixels; p++) {
        float c[3];
        for (int k = 0; k < 3; k++) { c[k] = rgb[p * 3 + k] / divisor; if (gamma) c[k] = sqrtf(c[k]); }
        bgra[p * 4 + 0] = quant(c[2]); bgra[p * 4 + 1] = quant(c[1]); bgra[p * 4 + 2] = quant(c[0]); bgra[p * 4 + 3] = 255;
    }
}

/* ---- unit-level entry points for the known-answer tests -------------------------------- */
int orc_ray_sphere(const float* o, const float* d, const float* c, float r, float tmin, float tmax, float* t) {
    ray_t ray; ray.o = pv(o); ray.d = pv(d); return ray_sphere(&ray, pv(c), r, tmin, tmax, t);
}
int orc_ray_plane(const float* o, const float* d, const float* n, const float* p0, float tmin, float tmax, float* t) {
    ray_t ray; ray.o = pv(o); ray.d = pv(d); return ray_plane(&ray, pv(n), pv(p0), tmin, tmax, t);
}
int orc_ray_rect_xy(const float* o, const float* d, float z, float x0, float x1, float y0, float y1, float tmin, float tmax, float* t) {
    ray_t ray; ray.o = pv(o); ray.d = pv(d); return ray_rect_xy(&ray, z, x0, x1, y0, y1, tmin, tmax, t);
}
int orc_ray_triangle(const float* o, const float* d, const float* v0, const float* v1, const float* v2,
                     float tmin, float tmax, float* t, float* bary) {
    ray_t ray; ray.o = pv(o); ray.d = pv(d); return ray_triangle(&ray, pv(v0), pv(v1), pv(v2), tmin, tmax, t, bary);
}
void orc_normalize(const float* v, float* out) { v3 r = normalize3(pv(v)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_reflect(const float* v, const float* n, float* out) { v3 r = reflect3(pv(v), pv(n)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_refract(const float* v, const float* n, float eta, int tir, float* out) {
    v3 r = refract3(pv(v), pv(n), eta, tir); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_rotation_mat(const float* deg, float* out9) { m3 r = rotation_mat(pv(deg)); memcpy(out9, r.m, sizeof r.m); }
float orc_reflectance(const float* v, const float* n, float ratio) { return reflectance(pv(v), pv(n), ratio); }

/* One prim query in path form: out = {t, px,py,pz, nx,ny,nz, front}. */
int orc_prim_hit_path(const rt_prim* p, const float* o, const float* d, float tmin, float tmax, float* out8) {
    ray_t ray; ray.o = pv(o); ray.d = pv(d);
    hit_t h; memset(&h, 0, sizeof h);
    int r = prim_hit_path(p, &ray, tmin, tmax, &h);
    if (r) { out8[0] = h.t; out8[1] = h.point.x; out8[2] = h.point.y; out8[3] = h.point.z;
             out8[4] = h.normal.x; out8[5] = h.normal.y; out8[6] = h.normal.z; out8[7] = (float)h.front; }
    return r;
}
/* Whole-scene closest hit in path form; returns the index semantics through out8 and *mat. */
int orc_scene_hit_path(const rt_scene* s, const float* o, const float* d, float tmin, float tmax, float* out8, int* mat) {
    ray_t ray; ray.o = pv(o); ray.d = pv(d);
    hit_t h; memset(&h, 0, sizeof h);
    int r = scene_hit_path(s, &ray, tmin, tmax, &h);
    if (r) { out8[0] = h.t; out8[1] = h.point.x; out8[2] = h.point.y; out8[3] = h.point.z;
             out8[4] = h.normal.x; out8[5] = h.normal.y; out8[6] = h.normal.z; out8[7] = (float)h.front; if (mat) *mat = h.material; }
    return r;
}
/* One scatter with explicit uniforms (u_fi, u_theta, u_schlick) instead of the counter RNG. */
void orc_scatter_explicit(int variant, const rt_material* m, const float* in_d, const float* point,
                          const float* normal, int front, float u0, float u1, float u2,
                          float* out_dir, float* out_att) {
    v3 dir, att, n = pv(normal), d = pv(in_d);
    ray_t r;
    switch (m->type) {
    case RT_MAT_DIFFUSE:
        dir = variant == RT_PATH_SKY ? rnd_hemisphere_03(n, 90.0f, u0, u1) : rnd_hemisphere_04(n, 90.0f, u0, u1);
        r = make_ray(pv(point), dir);
        if (variant == RT_PATH_SKY) att = pv(m->albedo);
        else { float a = dot3(r.d, n); if (!(0.0f < a)) a = 0.0f; att = muls(pv(m->albedo), a); }
        break;
    case RT_MAT_METAL:
        dir = reflect3(d, n);
        if (variant == RT_PATH_SKY) { if (m->roughness > 0.0f) dir = rnd_hemisphere_03(dir, 60.0f * m->roughness, u0, u1); }
        else dir = rnd_hemisphere_04(dir, 60.0f * m->roughness, u0, u1);
        r = make_ray(pv(point), dir);
        if (variant == RT_PATH_SKY) att = pv(m->albedo);
        else { float a = dot3(r.d, n); if (!(0.0f < a)) a = 0.0f; att = muls(pv(m->albedo), a); }
        break;
    case RT_MAT_DIELECTRIC: {
        float idx = front ? m->refraction_index : 1.0f / m->refraction_index;
        dir = reflect3(d, n);
        if (reflectance(d, n, 1.0f / idx) < u2) dir = refract3(d, n, idx, 1);
        r = make_ray(pv(point), dir);
        att = V(1, 1, 1);
        break;
    }
    default:
        r = make_ray(V(0, 0, 0), V(0, 0, -1.0f)); att = V(0, 0, 0);
    }
    out_dir[0] = r.d.x; out_dir[1] = r.d.y; out_dir[2] = r.d.z;
    out_att[0] = att.x; out_att[1] = att.y; out_att[2] = att.z;
}
void orc_mix(const float* a, const float* b, float r, float* out) { v3 m = mix3(pv(a), pv(b), r); out[0] = m.x; out[1] = m.y; out[2] = m.z; }
float orc_clamp(float v, float lo, float hi) { return clampf(v, lo, hi); }
void orc_ray_ctor(const float* d, float* out) { ray_t r = make_ray(V(0, 0, 0), pv(d)); out[0] = r.d.x; out[1] = r.d.y; out[2] = r.d.z; }
/* One prim query in Whitted (01/02) form: t and the raw normal. */
int orc_prim_hit_whitted(const rt_prim* p, const float* o, const float* d, float tmin, float tmax, float* t, float* n) {
    ray_t ray; ray.o = pv(o); ray.d = pv(d);
    v3 nn = V(0, 0, 0);
    int r = prim_hit_whitted(p, &ray, tmin, tmax, t, &nn);
    if (r) { n[0] = nn.x; n[1] = nn.y; n[2] = nn.z; }
    return r;
}
