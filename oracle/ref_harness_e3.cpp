// ref_harness_e3.cpp - the REFERENCE's 03_PathTracingBasics headers (Utils.h, Materials.hpp, Sphere.hpp),
// included from /root/reference where they lie.  TEST INFRASTRUCTURE ONLY (see ref_harness_e1.cpp).
// The reference's samplers own clock-seeded static engines, so sampler outputs are recorded as
// statistics (moments/histograms), not as exact vectors.
#include <cstring>
#include <cstdint>
#include <algorithm>
#include <Math.hpp>
#include <Ray.hpp>
#include "Utils.h"
#include "Materials.hpp"
#include "Sphere.hpp"

using V3 = math::Vec3<float>;
static V3 v(const float* p) { return V3(p[0], p[1], p[2]); }
static void st(float* o, const V3& a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; }
static math::Ray raw_ray(const float* o, const float* d) { math::Ray r; r.setOrigin(v(o)); r.setDirection(v(d), false); return r; }

extern "C" {
// scene = n spheres {cx,cy,cz,r,inverted}; closest hit through Scene::intersectsRay. out8 = t, point, normal, frontFace; *which = sphere index
int ref_e3_scene_hit(const float* spheres, int n, const float* o, const float* d, float tmin, float tmax, float* out8, int* which) {
    Scene sc;
    std::vector<std::shared_ptr<Material>> mats;
    for (int i = 0; i < n; ++i) {
        mats.push_back(std::make_shared<MaterialDiffuse>(V3((float)i, 0, 0)));
        sc.addElement(std::make_shared<Sphere>(v(spheres + 5 * i), spheres[5 * i + 3], mats.back(), spheres[5 * i + 4] != 0.0f));
    }
    HitInfo h{};
    int r = sc.intersectsRay(raw_ray(o, d), tmin, tmax, &h);
    if (r) {
        out8[0] = h.t; st(out8 + 1, h.point); st(out8 + 4, h.normal); out8[7] = h.frontFaceSurface ? 1.0f : 0.0f;
        *which = (int)static_cast<MaterialDiffuse*>(h.materialPtr.get())->getAlbedo().x;
    }
    return r;
}
// n draws of RndHemisphereVec(dir, thetaMax) -> out[3n]
void ref_e3_rnd_hemisphere(const float* dir, float theta_max, int n, float* out) {
    for (int i = 0; i < n; ++i) st(out + 3 * i, RndHemisphereVec(v(dir), theta_max));
}
// n scatters of one material at one hit: kind 1 diffuse(albedo), 2 metal(albedo, rough), 3 dielectric(ri). out dirs[3n] (normalised by setDirection), att[3]
void ref_e3_scatter(int kind, const float* albedo, float param, const float* in_d, const float* point, const float* normal, int front,
                    int n, float* out_dirs, float* out_att) {
    std::shared_ptr<Material> m;
    if (kind == 1) m = std::make_shared<MaterialDiffuse>(v(albedo));
    else if (kind == 2) m = std::make_shared<MaterialMetal>(v(albedo), param);
    else m = std::make_shared<MaterialDielectric>(param);
    HitInfo h{}; h.point = v(point); h.normal = v(normal); h.frontFaceSurface = front != 0; h.t = 1.0f;
    math::Ray in = raw_ray(point, in_d);
    for (int i = 0; i < n; ++i) {
        math::Ray sr; V3 att;
        m->scatter(in, h, &att, &sr);
        st(out_dirs + 3 * i, sr.getDirection());
        if (i == 0) st(out_att, att);
    }
}
}

// ---- converged-image golden ------------------------------------------------------------------------------------------
// Main.cpp cannot be built here (<windows.h>), so Render/TraceTay of 03 are driven from this file: the loop follows
// 03_PathTracingBasics/Main.cpp:327-349 (per-sample ray) and :380-435 (TraceTay) step for step, and everything it calls is
// the reference's own code - RndFloat, math::GetRotationMat, math::Ray, Scene::intersectsRay, Material::scatter, math::Mix -
// with the reference's own clock-seeded engines.  Constants of Main.cpp:13-17, :392.
namespace golden3 {
const unsigned kMaxDepth = 24;         // MAX_RECURSION_DEPTH
const unsigned kSamplesPerRay = 1;     // SAMPLES_PER_RAY
unsigned long long g_rays = 0;
bool trace(const math::Ray& ray, const HittableElement& world, V3* out, unsigned depth) {
    if (depth > kMaxDepth) { *out = {0.0f, 0.0f, 0.0f}; return false; }
    HitInfo hit{};
    ++g_rays;
    if (world.intersectsRay(ray, 0.001f, 1000.0f, &hit)) {
        if (hit.materialPtr != nullptr) {
            V3 result = {0.0f, 0.0f, 0.0f};
            for (unsigned s = 0; s < kSamplesPerRay; s++) {
                math::Ray scattered = {};
                V3 att = {};
                if (hit.materialPtr->scatter(ray, hit, &att, &scattered)) {
                    V3 child{};
                    trace(scattered, world, &child, depth + 1);
                    result = result + (att * child);
                }
            }
            *out = result / static_cast<float>(kSamplesPerRay);
        } else *out = {0.0f, 0.0f, 0.0f};
        return true;
    }
    auto k = 0.5 * (ray.getDirection().y + 1.0);
    *out = math::Mix(V3(1.0f, 1.0f, 1.0f), V3(0.5f, 0.7f, 1.0f), k);
    return false;
}
}  // namespace golden3

extern "C" {
// The committed scene and camera of 03 (Main.cpp:172-189) at w x h, rows [y0, y1); variant 1 = the hollow glass sphere of
// the commented-out lines :182-183 in place of the left metal sphere (the README screenshot's scene).
void ref_e3_golden_rows(int w, int h, int spp, int y0, int y1, int variant, double* sum, double* sumsq, unsigned long long* rays) {
    auto ground = std::make_shared<MaterialDiffuse>(V3(0.8f, 0.8f, 0.0f));
    auto center = std::make_shared<MaterialDiffuse>(V3(0.1f, 0.2f, 0.5f));
    auto left = std::make_shared<MaterialMetal>(V3(0.8f, 0.8f, 0.8f), 0.3f);
    auto right = std::make_shared<MaterialMetal>(V3(0.8f, 0.6f, 0.2f), 0.4f);
    auto glass = std::make_shared<MaterialDielectric>(0.625f);
    Scene world{};
    world.addElement(std::make_shared<Sphere>(V3(0, -100.5f, -1.0), 100.0f, ground));
    world.addElement(std::make_shared<Sphere>(V3(0.0f, 0.0f, -1.0f), 0.5f, center));
    if (variant == 1) {
        world.addElement(std::make_shared<Sphere>(V3(-1.0f, 0.0f, -1.0f), 0.5f, glass));
        world.addElement(std::make_shared<Sphere>(V3(-1.0f, 0.0f, -1.0f), 0.4f, glass, true));
    } else world.addElement(std::make_shared<Sphere>(V3(-1.0f, 0.0f, -1.0f), 0.5f, left));
    world.addElement(std::make_shared<Sphere>(V3(1.0f, 0.0f, -1.0f), 0.5f, right));
    const float fov = 50.0f;
    const V3 camPosition = {-2.0f, 1.5f, 1.0f}, camOrientation = {-30.0f, -42.0f, 0.0f};
    const float fw = static_cast<float>(w), fh = static_cast<float>(h);
    const float fovRadians = static_cast<float>(fov / (180.0f / M_PI));
    const unsigned samples = static_cast<unsigned>(spp);
    golden3::g_rays = 0;
    for (int j = y0; j < y1; j++) for (int i = 0; i < w; i++) {
        double* ps = sum + (static_cast<size_t>(j - y0) * w + i) * 3;
        double* pq = sumsq + (static_cast<size_t>(j - y0) * w + i) * 3;
        for (unsigned s = 0; s < samples; s++) {
            math::Vec2<float> bias = (samples > 1 ? math::Vec2<float>(RndFloat(), RndFloat()) : math::Vec2<float>(0.5f, 0.5f));
            float x = (2.0f * (static_cast<float>(i) + bias.x) / fw - 1.0f) * tanf(fovRadians / 2.0f) * fw / fh;
            float y = -(2.0f * (static_cast<float>(j) + bias.y) / fh - 1.0f) * tanf(fovRadians / 2.0f);
            V3 dir = math::GetRotationMat(camOrientation) * V3(x, y, -1.0f);
            math::Ray ray(camPosition, dir);
            V3 c = {0.0f, 0.0f, 0.0f};
            golden3::trace(ray, world, &c, 0);
            ps[0] += c.r; ps[1] += c.g; ps[2] += c.b;
            pq[0] += double(c.r) * c.r; pq[1] += double(c.g) * c.g; pq[2] += double(c.b) * c.b;
        }
    }
    *rays = golden3::g_rays;
}
}
