// ref_harness_e1.cpp - calls the REFERENCE's own header-only code (Common/Math.hpp, Common/Ray.hpp and the
// 01_Basic scene headers, included from /root/reference where they lie; nothing is copied) so that
// oracle/gen_golden.py can record known-answer vectors.  TEST INFRASTRUCTURE ONLY.
// Main.cpp (Render/TraceTay) is not built: it needs <windows.h>, which this image lacks.
#include <cstring>
#include <cstdint>
#include <algorithm>
#include <Math.hpp>
#include <Ray.hpp>
#include "Types.h"
#include "Sphere.hpp"
#include "Plane.hpp"

using V3 = math::Vec3<float>;
static V3 v(const float* p) { return V3(p[0], p[1], p[2]); }
static void st(float* o, const V3& a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; }

extern "C" {
// Ray built WITHOUT normalising (setDirection(dir,false)) so the raw routines see exactly the given vectors.
static math::Ray raw_ray(const float* o, const float* d) { math::Ray r; r.setOrigin(v(o)); r.setDirection(v(d), false); return r; }

int ref_ray_sphere(const float* o, const float* d, const float* c, float r, float tmin, float tmax, float* t) {
    return raw_ray(o, d).intersectsSphere(v(c), r, tmin, tmax, t);
}
int ref_ray_plane(const float* o, const float* d, const float* n, const float* p0, float tmin, float tmax, float* t) {
    return raw_ray(o, d).intersectsPlane(v(n), v(p0), tmin, tmax, t);
}
int ref_ray_rect_xy(const float* o, const float* d, float z, float x0, float x1, float y0, float y1, float tmin, float tmax, float* t) {
    return raw_ray(o, d).intersectsAARectangleXy(z, x0, x1, y0, y1, tmin, tmax, t);
}
int ref_ray_triangle(const float* o, const float* d, const float* v0, const float* v1, const float* v2, float tmin, float tmax, float* t, float* bary) {
    math::Vec2<float> b(0, 0);
    int r = raw_ray(o, d).intersectsTriangle(v(v0), v(v1), v(v2), tmin, tmax, t, &b);
    if (r && bary) { bary[0] = b.x; bary[1] = b.y; }
    return r;
}
void ref_normalize(const float* a, float* out) { st(out, math::Normalize(v(a))); }
void ref_reflect(const float* a, const float* n, float* out) { st(out, math::Reflect(v(a), v(n))); }
void ref_refract(const float* a, const float* n, float eta, int tir, float* out) { st(out, math::Refract(v(a), v(n), eta, tir != 0)); }
void ref_rotation_mat(const float* deg, float* out9) { auto m = math::GetRotationMat(v(deg)); std::memcpy(out9, m.data, sizeof(float) * 9); }
void ref_mix(const float* a, const float* b, float r, float* out) { st(out, math::Mix(v(a), v(b), r)); }
float ref_clamp(float x, float lo, float hi) { return math::Clamp(x, lo, hi); }
void ref_ray_ctor(const float* o, const float* d, float* out_d) { math::Ray r(v(o), v(d)); st(out_d, r.getDirection()); }

// 01/02 scene elements: Sphere/Plane::intersectsRay -> t and (not face-forwarded) normal
int ref_e1_sphere_hit(const float* c, float radius, const float* o, const float* d, float tmin, float tmax, float* t, float* n) {
    Sphere s(Material{}, v(c), radius); V3 nn; int r = s.intersectsRay(raw_ray(o, d), tmin, tmax, t, &nn); if (r) st(n, nn); return r;
}
int ref_e1_plane_hit(const float* p, const float* normal, const float* o, const float* d, float tmin, float tmax, float* t, float* n) {
    Plane s(Material{}, v(p), v(normal)); V3 nn; int r = s.intersectsRay(raw_ray(o, d), tmin, tmax, t, &nn); if (r) st(n, nn); return r;
}
}
