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

// ---- TraceTay on the host (host/03_PathTracingBasics/TraceTay.hpp) ---------------------------------------------------------------
namespace {
std::shared_ptr<Material> material_of(const rt_scene* sc, int idx) {
    if (idx < 0) return nullptr;
    const rt_material& m = sc->materials[idx];
    if (m.type == RT_MAT_DIFFUSE) return std::make_shared<MaterialDiffuse>(v(m.albedo));
    if (m.type == RT_MAT_METAL) return std::make_shared<MaterialMetal>(v(m.albedo), m.roughness);
    return std::make_shared<MaterialDielectric>(m.refraction_index);
}
bool build(const rt_scene* sc, Scene* world) {
    std::vector<std::shared_ptr<Material>> mats;
    for (int i = 0; i < sc->n_materials; ++i) mats.push_back(material_of(sc, i));
    for (int i = 0; i < sc->n_prims; ++i) {
        const rt_prim& p = sc->prims[i];
        if (p.type != RT_PRIM_SPHERE) return false;
        world->addElement(std::make_shared<Sphere>(v(p.f), p.f[3], p.material < 0 ? nullptr : mats[p.material], (p.flags & RT_PRIM_FLAG_FLIP_NORMALS) != 0));
    }
    return true;
}
}
// one ray (direction normalised by the Ray constructor, as every ray the reference traces): out4 = colour, hit
extern "C" int hm_e3_tracetay(const rt_scene* sc, const float* o, const float* d, unsigned depth, unsigned max_depth, float* out4) {
    Scene world; if (!build(sc, &world)) return 1;
    TraceTayMaxDepth() = max_depth;
    V3 c{-1.0f, -1.0f, -1.0f};
    out4[3] = TraceTay(math::Ray(v(o), v(d)), world, &c, depth) ? 1.0f : 0.0f;
    out4[0] = c.x; out4[1] = c.y; out4[2] = c.z;
    TraceTayMaxDepth() = 24u;
    return 0;
}
// the frame as the reference's Render samples it (Sources/03_PathTracingBasics/Main.cpp:327-349): mean linear colour per pixel
extern "C" int hm_e3_tracetay_frame(const rt_scene* sc, const rt_camera* cam, int w, int h, int spp, float* rgb) {
    Scene world; if (!build(sc, &world)) return 1;
    const float fw = static_cast<float>(w), fh = static_cast<float>(h);
    const float fovRadians = static_cast<float>(cam->fov_deg / (180.0f / M_PI));
    const auto rot = math::GetRotationMat(v(cam->orientation_deg));
    const unsigned samples = static_cast<unsigned>(spp);
    for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
        double acc[3] = {0, 0, 0};
        for (unsigned s = 0; s < samples; ++s) {
            const math::Vec2<float> bias = samples > 1 ? math::Vec2<float>(RndFloat(), RndFloat()) : math::Vec2<float>(0.5f, 0.5f);
            const float x = (2.0f * (static_cast<float>(i) + bias.x) / fw - 1.0f) * tanf(fovRadians / 2.0f) * fw / fh;
            const float y = -(2.0f * (static_cast<float>(j) + bias.y) / fh - 1.0f) * tanf(fovRadians / 2.0f);
            V3 c{0.0f, 0.0f, 0.0f};
            TraceTay(math::Ray(v(cam->position), rot * V3(x, y, -1.0f)), world, &c, 0);
            acc[0] += c.x; acc[1] += c.y; acc[2] += c.z;
        }
        float* o = rgb + 3 * (static_cast<size_t>(j) * w + i);
        for (int k = 0; k < 3; ++k) o[k] = static_cast<float>(acc[k] / samples);
    }
    return 0;
}
