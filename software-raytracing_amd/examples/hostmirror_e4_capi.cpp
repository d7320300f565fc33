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
    math::Ray scatteredRay(const math::Ray&, const HitInfo&, V3*) const override { return {}; }
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

// n scatters of one host material at one hit (the reference interface: isScatters / isEmits / scatteredRay / emittedColor).
// kind 1 diffuse, 2 metal(rough), 3 refractive(index), 4 light.  out: dirs[3n], atts[3n], flags[2], emitted[3]
extern "C" void hm_e4_scatter(int kind, const float* albedo, float param, const float* in_d, const float* point, const float* normal, int front,
                              int n, float* out_dirs, float* out_atts, int* flags, float* emitted) {
    std::shared_ptr<materials::Material> m;
    if (kind == 1) m = std::make_shared<materials::Diffuse>(v(albedo));
    else if (kind == 2) m = std::make_shared<materials::Metal>(v(albedo), param);
    else if (kind == 3) m = std::make_shared<materials::Refractive>(param);
    else m = std::make_shared<materials::Light>(v(albedo));
    HitInfo h{}; h.point = v(point); h.normal = v(normal); h.frontFaceSurface = front != 0; h.t = 1.0f;
    math::Ray in; in.setOrigin(v(point)); in.setDirection(v(in_d), false);
    flags[0] = m->isScatters(in, h); flags[1] = m->isEmits(in, h);
    const V3 e = m->emittedColor(); emitted[0] = e.x; emitted[1] = e.y; emitted[2] = e.z;
    for (int i = 0; i < n; ++i) {
        V3 att(0, 0, 0);
        const math::Ray sr = m->scatteredRay(in, h, &att);
        const V3 d = sr.getDirection();
        out_dirs[3 * i] = d.x; out_dirs[3 * i + 1] = d.y; out_dirs[3 * i + 2] = d.z;
        out_atts[3 * i] = att.x; out_atts[3 * i + 1] = att.y; out_atts[3 * i + 2] = att.z;
    }
}
// A nested list and a triangle element, flattened as Render() does: returns the number of flat primitives and their types.
extern "C" int hm_e4_flatten_nested(int* types, int cap) {
    auto white = std::make_shared<materials::Diffuse>(V3(0.7f, 0.7f, 0.7f));
    auto inner = std::make_shared<scene::List>();
    inner->addElement(std::make_shared<scene::Sphere>(white, V3(0, 0, -3), 1.0f));
    inner->addElement(std::make_shared<scene::Triangle>(white, V3(-1, 0, -2), V3(1, 0, -2), V3(0, 1, -2)));
    auto deeper = std::make_shared<scene::List>(std::make_shared<scene::Plane>(white, V3(0, -1, 0), V3(0, 1, 0)));
    inner->addElement(deeper);
    scene::List world;
    world.addElement(std::make_shared<scene::Plane>(white, V3(0, 5, 0), V3(0, -1, 0)));
    world.addElement(inner);
    world.addElement(std::make_shared<scene::Rectangle>(white, V3(0, 4, 0), math::Vec2<float>(1, 1), V3(90, 0, 0)));
    rtbackend::Flat flat;
    path_detail::flatten(world.getElements(), &flat);
    for (size_t i = 0; i < flat.prims.size() && static_cast<int>(i) < cap; ++i) types[i] = flat.prims[i].type;
    return static_cast<int>(flat.prims.size()) * 100 + static_cast<int>(flat.mats.size());
}
// host Triangle element against the raw routine (same t; normal turned against the ray)
extern "C" int hm_e4_triangle_hit(const float* v0, const float* v1, const float* v2, const float* o, const float* d, float tmin, float tmax, float* out8) {
    scene::Triangle tri(nullptr, v(v0), v(v1), v(v2));
    math::Ray ray; ray.setOrigin(v(o)); ray.setDirection(v(d), false);
    HitInfo h{};
    const bool hit = tri.intersectsRay(ray, tmin, tmax, &h);
    if (hit) { out8[0] = h.t; out8[1] = h.point.x; out8[2] = h.point.y; out8[3] = h.point.z; out8[4] = h.normal.x; out8[5] = h.normal.y; out8[6] = h.normal.z; out8[7] = h.frontFaceSurface ? 1.0f : 0.0f; }
    return hit;
}

// ---- TraceTay on the host (host/04_PathTracingLights/TraceTay.hpp) ---------------------------------------------------------------
namespace {
bool build(const rt_scene* sc, scene::List* world) {
    std::vector<std::shared_ptr<materials::Material>> mats;
    for (int i = 0; i < sc->n_materials; ++i) {
        const rt_material& m = sc->materials[i];
        if (m.type == RT_MAT_DIFFUSE) mats.push_back(std::make_shared<materials::Diffuse>(v(m.albedo)));
        else if (m.type == RT_MAT_METAL) mats.push_back(std::make_shared<materials::Metal>(v(m.albedo), m.roughness));
        else if (m.type == RT_MAT_DIELECTRIC) mats.push_back(std::make_shared<materials::Refractive>(m.refraction_index));
        else mats.push_back(std::make_shared<materials::Light>(v(m.albedo)));
    }
    for (int i = 0; i < sc->n_prims; ++i) {
        const rt_prim& p = sc->prims[i]; const float* f = p.f;
        const std::shared_ptr<materials::Material> m = p.material < 0 ? nullptr : mats[p.material];
        const bool flip = (p.flags & RT_PRIM_FLAG_FLIP_NORMALS) != 0;
        if (p.type == RT_PRIM_SPHERE) world->addElement(std::make_shared<scene::Sphere>(m, v(f), f[3], flip));
        else if (p.type == RT_PRIM_PLANE) world->addElement(std::make_shared<scene::Plane>(m, v(f), v(f + 3)));
        else if (p.type == RT_PRIM_RECT) world->addElement(std::make_shared<scene::Rectangle>(m, v(f), math::Vec2<float>(f[3], f[4]), v(f + 5)));
        else if (p.type == RT_PRIM_BOX) world->addElement(std::make_shared<scene::Box>(m, v(f), v(f + 3), v(f + 6), flip));
        else if (p.type == RT_PRIM_TRIANGLE) world->addElement(std::make_shared<scene::Triangle>(m, v(f), v(f + 3), v(f + 6)));
        else return false;
    }
    return true;
}
}
// one ray (direction normalised by the Ray constructor): out4 = colour, hit
extern "C" int hm_e4_tracetay(const rt_scene* sc, const float* o, const float* d, unsigned depth, unsigned max_depth, float* out4) {
    scene::List world; if (!build(sc, &world)) return 1;
    TraceTayMaxDepth() = max_depth;
    V3 c{-1.0f, -1.0f, -1.0f};
    out4[3] = TraceTay(math::Ray(v(o), v(d)), world, &c, depth) ? 1.0f : 0.0f;
    out4[0] = c.x; out4[1] = c.y; out4[2] = c.z;
    TraceTayMaxDepth() = 6u;
    return 0;
}
// the frame as the reference's Render samples it (Sources/04_PathTracingLights/Main.cpp:355-377): mean linear colour per pixel
extern "C" int hm_e4_tracetay_frame(const rt_scene* sc, const rt_camera* cam, int w, int h, int spp, float* rgb) {
    scene::List world; if (!build(sc, &world)) return 1;
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
