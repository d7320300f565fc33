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

// ---- converged-image golden ------------------------------------------------------------------------------------------
// Main.cpp cannot be built here (<windows.h>), so Render/TraceTay of 04 are driven from this file: the loop below follows
// 04_PathTracingLights/Main.cpp:355-377 (per-sample ray) and :431-502 (TraceTay) step for step, but EVERYTHING it calls is the
// reference's own code - RndFloat, math::GetRotationMat, math::Ray, scene::List::intersectsRay, Material::isScatters /
// scatteredRay / isEmits / emittedColor, with the reference's own <random> engines.  Constants of Main.cpp:20-26, :447.
namespace golden4 {
const unsigned kMaxDepth = 6;          // MAX_RECURSION_DEPTH
const unsigned kSamplesPerRay = 1;     // SAMPLES_PER_RAY
unsigned long long g_rays = 0;
bool trace(const math::Ray& ray, const scene::Hittable& world, V3* out, unsigned depth) {
    if (depth > kMaxDepth) { *out = {0.0f, 0.0f, 0.0f}; return false; }
    HitInfo hit{};
    ++g_rays;
    if (world.intersectsRay(ray, 0.01f, 1000.0f, &hit)) {
        if (hit.materialPtr != nullptr) {
            V3 result = {0.0f, 0.0f, 0.0f};
            if (hit.materialPtr->isScatters(ray, hit)) {
                for (unsigned s = 0; s < kSamplesPerRay; s++) {
                    V3 att = {0.0f, 0.0f, 0.0f};
                    auto scattered = hit.materialPtr->scatteredRay(ray, hit, &att);
                    V3 child = {0.0f, 0.0f, 0.0f};
                    trace(scattered, world, &child, depth + 1);
                    result = result + (att * child);
                }
                result = result / static_cast<float>(kSamplesPerRay);
            }
            if (hit.materialPtr->isEmits(ray, hit)) result = result + hit.materialPtr->emittedColor();
            *out = result;
        } else *out = {0.0f, 0.0f, 0.0f};
        return true;
    }
    *out = {0.0f, 0.0f, 0.0f};
    return false;
}
}  // namespace golden4

extern "C" {
// The committed scene and camera of 04 (Main.cpp:181-203) at w x h, rows [y0, y1), 'spp' samples per pixel.
// sum / sumsq: per pixel and channel, sum of the sample colours and of their squares (double), (y1-y0)*w*3 each.
void ref_e4_golden_rows(int w, int h, int spp, int y0, int y1, double* sum, double* sumsq, unsigned long long* rays) {
    auto red = std::make_shared<materials::Diffuse>(V3(0.65f, 0.05f, 0.05f));
    auto white = std::make_shared<materials::Diffuse>(V3(0.73f, 0.73f, 0.73f));
    auto green = std::make_shared<materials::Diffuse>(V3(0.12f, 0.45f, 0.15f));
    auto light = std::make_shared<materials::Light>(V3(15.0f, 15.0f, 15.0f));
    auto metal = std::make_shared<materials::Metal>(V3(0.8f, 0.8f, 0.8f), 0.2f);
    auto ball = std::make_shared<materials::Diffuse>(V3(0.1f, 0.2f, 0.5f));
    auto glass = std::make_shared<materials::Refractive>(0.6f);
    scene::List world{};
    world.addElement(std::make_shared<scene::Plane>(white, V3(0.0f, 5.0f, 0.0f), V3(0.0f, -1.0f, 0.0f)));
    world.addElement(std::make_shared<scene::Plane>(white, V3(0.0f, -5.0f, 0.0f), V3(0.0f, 1.0f, 0.0f)));
    world.addElement(std::make_shared<scene::Plane>(white, V3(0.0f, 0.0f, -5.0f), V3(0.0f, 0.0f, 1.0f)));
    world.addElement(std::make_shared<scene::Plane>(red, V3(-5.0f, 0.0f, 0.0f), V3(1.0f, 0.0f, 0.0f)));
    world.addElement(std::make_shared<scene::Plane>(green, V3(5.0f, 0.0f, 0.0f), V3(-1.0f, 0.0f, 0.0f)));
    world.addElement(std::make_shared<scene::Sphere>(metal, V3(0.0f, -3.0f, -1.0), 2.0f));
    world.addElement(std::make_shared<scene::Sphere>(ball, V3(-2.0f, -4.0f, 2.5), 1.0f));
    world.addElement(std::make_shared<scene::Sphere>(glass, V3(2.5f, -3.5f, 3.0f), 1.5f));
    world.addElement(std::make_shared<scene::Rectangle>(light, V3(0.0f, 4.95f, 0.0f), math::Vec2<float>(3.0f, 3.0f), V3(90.0f, 0.0f, 0.0f)));
    const float fov = 90.0f;
    const V3 viewPosition = {0.0f, 0.0f, 10.0f}, viewOrient = {0.0f, 0.0f, 0.0f};
    const float fw = static_cast<float>(w), fh = static_cast<float>(h);
    const float fovRadians = static_cast<float>(fov / (180.0f / M_PI));
    const unsigned samples = static_cast<unsigned>(spp);
    golden4::g_rays = 0;
    for (int row = y0; row < y1; row++) for (int col = 0; col < w; col++) {
        double* ps = sum + (static_cast<size_t>(row - y0) * w + col) * 3;
        double* pq = sumsq + (static_cast<size_t>(row - y0) * w + col) * 3;
        for (unsigned s = 0; s < samples; s++) {
            math::Vec2<float> bias = (samples > 1 ? math::Vec2<float>(RndFloat(), RndFloat()) : math::Vec2<float>(0.5f, 0.5f));
            float x = (2.0f * (static_cast<float>(col) + bias.x) / fw - 1.0f) * tanf(fovRadians / 2.0f) * fw / fh;
            float y = -(2.0f * (static_cast<float>(row) + bias.y) / fh - 1.0f) * tanf(fovRadians / 2.0f);
            V3 dir = math::GetRotationMat(viewOrient) * V3(x, y, -1.0f);
            math::Ray ray(viewPosition, dir);
            V3 c = {0.0f, 0.0f, 0.0f};
            golden4::trace(ray, world, &c, 0);
            ps[0] += c.r; ps[1] += c.g; ps[2] += c.b;
            pq[0] += double(c.r) * c.r; pq[1] += double(c.g) * c.g; pq[2] += double(c.b) * c.b;
        }
    }
    *rays = golden4::g_rays;
}
}
