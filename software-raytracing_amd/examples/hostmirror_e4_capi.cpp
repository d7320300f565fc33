// hostmirror_e4_capi.cpp - C wrapper around the host-side 04 scene classes (host/04_PathTracingLights) so that
// tests/test_host_mirror.py can check scene::List / Sphere / Plane / Rectangle / Box::intersectsRay bit for bit against
// the closest hits recorded from the reference's own classes (tests/golden/kat_scene_e4.npz, kat_box_e4.npz).
// Test support only; needs no GPU (nothing here calls Render).
#include "../host/04_PathTracingLights/Render.hpp"

namespace {
using V3 = math::Vec3<float>;
V3 v(const float* p) { return {p[0], p[1], p[2]}; }
// carries the element index back through HitInfo::materialPtr
struct Tag : materials::Material {
    int idx;
    explicit Tag(int i) : idx(i) {}
    bool isScatters(const math::Ray&, const HitInfo&) const override { return false; }
    bool isEmits(const math::Ray&, const HitInfo&) const override { return false; }
    V3 emittedColor() const override { return {}; }
    bool describe(rt_material*) const override { return false; }
};
}

// prims: n records of 13 floats {kind (0 sphere, 1 plane, 2 rectangle, 4 box), flip, f[0..10] in rt_prim order}
extern "C" int hm_e4_scene_hit(const float* prims, int n, const float* o, const float* d, float tmin, float tmax, float* out8, int* which) {
    scene::List list;
    for (int i = 0; i < n; ++i) {
        const float* p = prims + 13 * i; const float* f = p + 2;
        auto tag = std::make_shared<Tag>(i);
        const int kind = static_cast<int>(p[0]);
        if (kind == 0) list.addElement(std::make_shared<scene::Sphere>(tag, v(f), f[3], p[1] != 0.0f));
        else if (kind == 1) list.addElement(std::make_shared<scene::Plane>(tag, v(f), v(f + 3)));
        else if (kind == 4) list.addElement(std::make_shared<scene::Box>(tag, v(f), v(f + 3), v(f + 6), p[1] != 0.0f));
        else list.addElement(std::make_shared<scene::Rectangle>(tag, v(f), math::Vec2<float>(f[3], f[4]), v(f + 5)));
    }
    math::Ray ray; ray.setOrigin(v(o)); ray.setDirection(v(d), false);   // recorded directions are unit already
    HitInfo h{};
    const bool hit = list.intersectsRay(ray, tmin, tmax, &h);
    if (hit) {
        out8[0] = h.t; out8[1] = h.point.x; out8[2] = h.point.y; out8[3] = h.point.z;
        out8[4] = h.normal.x; out8[5] = h.normal.y; out8[6] = h.normal.z; out8[7] = h.frontFaceSurface ? 1.0f : 0.0f;
        *which = static_cast<Tag*>(h.materialPtr.get())->idx;
    }
    return hit;
}
