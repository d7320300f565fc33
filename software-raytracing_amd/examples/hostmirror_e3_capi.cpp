// hostmirror_e3_capi.cpp - C wrapper around the host-side 03 material classes (host/03_PathTracingBasics) so that
// tests/test_host_mirror.py can check Material::scatter() - the reference interface kept for callers of their own - against
// the statistics recorded from the reference's classes (tests/golden/stat_samplers_materials.npz).  Test support only; no GPU.
#include "../host/03_PathTracingBasics/Render.hpp"

namespace {
using V3 = math::Vec3<float>;
V3 v(const float* p) { return {p[0], p[1], p[2]}; }
}
// kind 1 diffuse(albedo), 2 metal(albedo, rough), 3 dielectric(index).  out: dirs[3n], att[3]; returns scatter()'s result of the last call
extern "C" int hm_e3_scatter(int kind, const float* albedo, float param, const float* in_d, const float* point, const float* normal, int front,
                             int n, float* out_dirs, float* out_att) {
    std::shared_ptr<Material> m;
    if (kind == 1) m = std::make_shared<MaterialDiffuse>(v(albedo));
    else if (kind == 2) m = std::make_shared<MaterialMetal>(v(albedo), param);
    else m = std::make_shared<MaterialDielectric>(param);
    HitInfo h{}; h.point = v(point); h.normal = v(normal); h.frontFaceSurface = front != 0; h.t = 1.0f;
    math::Ray in; in.setOrigin(v(point)); in.setDirection(v(in_d), false);
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        math::Ray sr; V3 att;
        ok = m->scatter(in, h, &att, &sr);
        const V3 d = sr.getDirection();
        out_dirs[3 * i] = d.x; out_dirs[3 * i + 1] = d.y; out_dirs[3 * i + 2] = d.z;
        if (i == 0) { out_att[0] = att.x; out_att[1] = att.y; out_att[2] = att.z; }
    }
    return ok;
}
// a Scene nested in a Scene flattens to its elements in order: returns the number of flat primitives
extern "C" int hm_e3_flatten_nested() {
    auto m = std::make_shared<MaterialDiffuse>(V3(0.5f, 0.5f, 0.5f));
    auto inner = std::make_shared<Scene>(std::make_shared<Sphere>(V3(0, 0, -1), 0.5f, m));
    inner->addElement(std::make_shared<Sphere>(V3(1, 0, -1), 0.5f, m));
    Scene world(std::make_shared<Sphere>(V3(0, -100.5f, -1), 100.0f, m));
    world.addElement(inner);
    rtbackend::Flat flat;
    path_detail::flatten(world.getElements(), &flat);
    return static_cast<int>(flat.prims.size());
}
