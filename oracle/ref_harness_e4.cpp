// ref_harness_e4.cpp - the REFERENCE's 04_PathTracingLights headers (Utils.h, Scene/*.hpp, Materials/*.hpp),
// included from /root/reference where they lie.  TEST INFRASTRUCTURE ONLY (see ref_harness_e1.cpp).
#include <cstring>
#include <cstdint>
#include <algorithm>
#include <Math.hpp>
#include <Ray.hpp>
#include "Utils.h"
#include "Scene/Sphere.hpp"
#include "Scene/Plane.hpp"
#include "Scene/Rectangle.hpp"
#include "Scene/Box.hpp"
#include "Materials/Diffuse.hpp"
#include "Materials/Metal.hpp"
#include "Materials/Refractive.hpp"
#include "Materials/Light.hpp"

using V3 = math::Vec3<float>;
static V3 v(const float* p) { return V3(p[0], p[1], p[2]); }
static void st(float* o, const V3& a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; }
static math::Ray raw_ray(const float* o, const float* d) { math::Ray r; r.setOrigin(v(o)); r.setDirection(v(d), false); return r; }

// a material whose only job is to carry the element index back through HitInfo::materialPtr
struct Tag : materials::Material {
    int idx; explicit Tag(int i) : idx(i) {}
    bool isScatters(const math::Ray&, const HitInfo&) const override { return false; }
    bool isEmits(const math::Ray&, const HitInfo&) const override { return false; }
    math::Ray scatteredRay(const math::Ray&, const HitInfo&, V3*) const override { return {}; }
    V3 emittedColor() const override { return {}; }
};

extern "C" {
// prims: n records of 13 floats {type(0 sphere,1 plane,2 rect,4 box), flip, f[0..10]} in rt_prim field order; closest hit via scene::List.
int ref_e4_scene_hit(const float* prims, int n, const float* o, const float* d, float tmin, float tmax, float* out8, int* which) {
    scene::List list;
    for (int i = 0; i < n; ++i) {
        const float* p = prims + 13 * i; const float* f = p + 2;
        auto tag = std::make_shared<Tag>(i);
        if (p[0] == 0.0f) list.addElement(std::make_shared<scene::Sphere>(tag, v(f), f[3], p[1] != 0.0f));
        else if (p[0] == 1.0f) list.addElement(std::make_shared<scene::Plane>(tag, v(f), v(f + 3)));
        else if (p[0] == 4.0f) list.addElement(std::make_shared<scene::Box>(tag, v(f), v(f + 3), v(f + 6), p[1] != 0.0f));
        else list.addElement(std::make_shared<scene::Rectangle>(tag, v(f), math::Vec2<float>(f[3], f[4]), v(f + 5)));
    }
    HitInfo h{};
    int r = list.intersectsRay(raw_ray(o, d), tmin, tmax, &h);
    if (r) {
        out8[0] = h.t; st(out8 + 1, h.point); st(out8 + 4, h.normal); out8[7] = h.frontFaceSurface ? 1.0f : 0.0f;
        *which = static_cast<Tag*>(h.materialPtr.get())->idx;
    }
    return r;
}
void ref_e4_rnd_hemisphere3(const float* dir, float theta_max, int n, float* out) {
    for (int i = 0; i < n; ++i) st(out + 3 * i, RndHemisphereVec3(v(dir), theta_max));
}
// kind 1 diffuse, 2 metal(rough), 3 refractive(ri), 4 light. out: dirs[3n], atts[3n], flags[2] = {isScatters, isEmits}, emitted[3]
void ref_e4_scatter(int kind, const float* albedo, float param, const float* in_d, const float* point, const float* normal, int front,
                    int n, float* out_dirs, float* out_atts, int* flags, float* emitted) {
    std::shared_ptr<materials::Material> m;
    if (kind == 1) m = std::make_shared<materials::Diffuse>(v(albedo));
    else if (kind == 2) m = std::make_shared<materials::Metal>(v(albedo), param);
    else if (kind == 3) m = std::make_shared<materials::Refractive>(param);
    else m = std::make_shared<materials::Light>(v(albedo));
    HitInfo h{}; h.point = v(point); h.normal = v(normal); h.frontFaceSurface = front != 0; h.t = 1.0f;
    math::Ray in = raw_ray(point, in_d);
    flags[0] = m->isScatters(in, h); flags[1] = m->isEmits(in, h);
    st(emitted, m->emittedColor());
    for (int i = 0; i < n; ++i) {
        V3 att(0, 0, 0);
        math::Ray sr = m->scatteredRay(in, h, &att);
        st(out_dirs + 3 * i, sr.getDirection()); st(out_atts + 3 * i, att);
    }
}
}
